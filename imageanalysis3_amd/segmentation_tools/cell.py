"""segmentation_tools/cell.py:598-611 — bounding box of a label mask (used by DaxProcesser._fit_spots_by_segmentation)."""
import numpy as np


def segmentation_mask_2_bounding_box(mask, cell_id=None, extend_pixel=1):
    """``ImageCrop_3d`` around the non-zero voxels of ``mask`` (or of ``mask == cell_id`` when that label occurs),
    grown by ``extend_pixel`` and clipped to the image."""
    from ..classes.preprocess import ImageCrop_3d
    if cell_id is not None and (mask == cell_id).any():
        _mask = (mask == cell_id)
    else:
        _mask = mask
    extend_pixel = int(extend_pixel)
    _crop = []
    for _i, _sz in enumerate(_mask.shape):
        _inds = np.where(np.max(_mask, axis=tuple(np.setdiff1d(np.arange(len(_mask.shape)), _i))))[0]
        _crop.append([max(np.min(_inds) - extend_pixel, 0),
                      min(np.max(_inds) + 1 + extend_pixel, _sz)])
    return ImageCrop_3d(_crop, _mask.shape)
