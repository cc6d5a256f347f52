"""TEST HARNESS (not part of the product package). Bounding box of a segmentation label (interface of the reference's segmentation_tools/cell.py:598-611; used by
DaxProcesser._fit_spots_by_segmentation)."""
import numpy as np


def segmentation_mask_2_bounding_box(mask, cell_id=None, extend_pixel=1):
    """``ImageCrop_3d`` around ``mask == cell_id`` when that label occurs, otherwise around the non-zero voxels of
    ``mask``; grown by ``extend_pixel`` on every side and clipped to the image."""
    from imageanalysis3_amd.classes.preprocess import ImageCrop_3d
    sel = np.asarray(mask)
    if cell_id is not None:
        labelled = sel == cell_id
        if labelled.any():
            sel = labelled
    pad = int(extend_pixel)
    hit = np.nonzero(sel)                      # one index vector per axis
    box = [[max(int(ix.min()) - pad, 0), min(int(ix.max()) + 1 + pad, n)] for ix, n in zip(hit, sel.shape)]
    return ImageCrop_3d(box, sel.shape)
