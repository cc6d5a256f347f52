"""TEST HARNESS (not part of the product package). Spot-to-label lookups the spot-calling path uses (interface of the reference's classes/partition_spots.py:113-157,
:212-236).  The gene-count tables, plots and file merging of ``Spots_Partition`` are downstream analysis, out of scope."""
import numpy as np
from imageanalysis3_amd.classes.preprocess import Spots3D  # noqa: F401  (the spot container these helpers take)

default_search_radius = 4
default_pixel_sizes = [250, 108, 108]


def find_coordinate_intensities(image, spots, search_radius=5):
    """:212-236 — ``image`` sampled on the (2r+1)³ cube around every spot centre (rounded to voxels, indices clamped
    to the image); one row of (2r+1)³ values per spot, cube offsets in the reference's meshgrid order."""
    shape = np.array(np.shape(image))
    centre = np.round(spots.to_coords()).astype(np.int32)                 # (n, 3)
    r = np.arange(-search_radius, search_radius + 1)
    # np.meshgrid's default 'xy' indexing swaps the first two axes; the reference flattens it as (2, 1, 3) -> rows
    offsets = np.stack(np.meshgrid(r, r, r)).transpose((2, 1, 3, 0)).reshape(-1, 3)
    idx = centre[None, :, :] + offsets[:, None, :]                         # (cube, n, 3)
    idx = np.clip(idx, 0, shape - 1)
    return np.asarray(image)[idx[..., 0], idx[..., 1], idx[..., 2]].T


class Spots_Partition():
    """Static helpers of the reference class (:113-157)."""

    @staticmethod
    def spots_to_labels(segmentation_masks, spots, search_radius=10, verbose=True):
        """Most frequent positive label in every spot's neighbourhood, -1 when there is none (ties: smallest label)."""
        if verbose:
            print(f"-- partition barcodes for {len(spots)} spots")
        labels = np.full(len(spots), -1, dtype=np.int32)
        for k, cube in enumerate(find_coordinate_intensities(segmentation_masks, spots, search_radius=search_radius)):
            values, counts = np.unique(cube[cube > 0], return_counts=True)
            if len(values):
                labels[k] = values[np.argmax(counts)]
        return labels

    @staticmethod
    def spots_to_DAPI(dapi_im, spots, search_radius=5, verbose=True):
        """Brightest DAPI voxel in every spot's neighbourhood."""
        if verbose:
            print(f"-- calculate local DAPI signal for {len(spots)} spots")
        return find_coordinate_intensities(dapi_im, spots, search_radius=search_radius).max(axis=1)
