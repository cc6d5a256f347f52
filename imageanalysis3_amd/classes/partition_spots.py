"""classes/partition_spots.py — the two spot-to-label helpers the spot-calling path uses (reference :113-157, :212-236).
The gene-count tables, plots and file merging of Spots_Partition are downstream analysis, out of scope."""
import numpy as np
from .preprocess import Spots3D

default_search_radius = 4
default_pixel_sizes = [250, 108, 108]


def find_coordinate_intensities(image, spots, search_radius=5):
    """:212-236 — values of ``image`` in the (2r+1)³ neighbourhood of every (rounded, border-clamped) spot centre;
    returns (n_spots, (2r+1)³)."""
    image_size = np.array(np.shape(image))
    _coords = np.round(spots.to_coords()).astype(np.int32).transpose()
    _r = np.arange(-search_radius, search_radius + 1)
    _local_coords = np.stack(np.meshgrid(_r, _r, _r)).transpose((2, 1, 3, 0)).reshape(-1, 3)
    all_ints = []
    for _lc in _local_coords:
        _modified_coords = _coords + _lc[:, np.newaxis]
        for _ic, _size in enumerate(image_size):
            _modified_coords[_ic][_modified_coords[_ic] < 0] = 0
            _modified_coords[_ic][_modified_coords[_ic] >= _size] = _size - 1
        all_ints.append(image[tuple(_modified_coords)])
    return np.array(all_ints).transpose()


class Spots_Partition():
    """Static helpers of the reference class (:113-157)."""

    @staticmethod
    def spots_to_labels(segmentation_masks, spots, search_radius=10, verbose=True):
        """Most frequent positive label around every spot, -1 when there is none."""
        if verbose:
            print(f"-- partition barcodes for {len(spots)} spots")
        _spot_labels = []
        _signals = find_coordinate_intensities(segmentation_masks, spots, search_radius=search_radius)
        for _spot_signal in _signals:
            _mks, _counts = np.unique(_spot_signal, return_counts=True)
            _counts = _counts[_mks > 0]
            _mks = _mks[_mks > 0]
            if len(_mks) == 0:
                _spot_labels.append(-1)
            else:
                _spot_labels.append(_mks[np.argmax(_counts)])
        return np.array(_spot_labels, dtype=np.int32)

    @staticmethod
    def spots_to_DAPI(dapi_im, spots, search_radius=5, verbose=True):
        if verbose:
            print(f"-- calculate local DAPI signal for {len(spots)} spots")
        return np.max(find_coordinate_intensities(dapi_im, spots, search_radius=search_radius), axis=1)
