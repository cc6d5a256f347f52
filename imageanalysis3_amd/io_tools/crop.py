"""reference: io_tools/crop.py:59-88."""
import numpy as np
from .. import _image_size


def generate_neighboring_crop(coord, crop_size=5, single_im_size=_image_size, sub_pixel_precision=False):
    """Box of +-crop_size around ``coord`` clipped to the image (io_tools/crop.py:59-88)."""
    from ..classes.preprocess import ImageCrop
    _coord = np.array(coord)[:len(single_im_size)]
    if isinstance(crop_size, (int, np.integer)):
        _crop_size = np.ones(len(single_im_size), dtype=np.int32) * crop_size
    else:
        _crop_size = np.array(crop_size)[:len(single_im_size)]
    _size = np.array(single_im_size, dtype=np.int32)
    if sub_pixel_precision:
        _left = np.max([_coord - _crop_size, np.zeros(len(_size))], axis=0)
        _right = np.min([_coord + _crop_size + 1, _size], axis=0)
    else:
        _left = np.max([np.round(_coord - _crop_size), np.zeros(len(_size))], axis=0)
        _right = np.min([np.round(_coord + _crop_size + 1), _size], axis=0)
    return ImageCrop(len(single_im_size), np.array([_left, _right]).transpose(), single_im_size=single_im_size)
