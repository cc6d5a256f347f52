"""Neighbourhood boxes around fitted spots (interface of the reference's io_tools/crop.py:59-88)."""
import numpy as np
from .. import _image_size


def generate_neighboring_crop(coord, crop_size=5, single_im_size=_image_size, sub_pixel_precision=False):
    """``ImageCrop`` covering ``coord - crop_size .. coord + crop_size`` (inclusive) on every axis, clipped to the
    image; limits are rounded to whole pixels unless ``sub_pixel_precision`` (the int32 box then truncates them)."""
    from ..classes.preprocess import ImageCrop
    size = np.asarray(single_im_size)
    nd = len(size)
    half = np.broadcast_to(np.asarray(crop_size), (nd,)) if np.ndim(crop_size) == 0 else np.asarray(crop_size)[:nd]
    centre = np.asarray(coord)[:nd]
    lo, hi = centre - half, centre + half + 1
    if not sub_pixel_precision:
        lo, hi = np.round(lo), np.round(hi)
    box = np.stack([np.maximum(lo, 0), np.minimum(hi, size.astype(np.int32))], axis=1)
    return ImageCrop(nd, box, single_im_size=single_im_size)
