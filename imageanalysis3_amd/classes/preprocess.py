"""Row schema and light containers around the spot table (reference: classes/preprocess.py:13-251).

Only what wraps the hot path's inputs/outputs is provided: the 11-column row naming, ``ImageCrop`` /
``ImageCrop_3d`` ([start, stop) boxes) and the ``Spots3D`` ndarray view.  ``DaxProcesser`` (the
step-by-step driver, :337-1254) is a caller of the operators, out of scope.
"""
import numpy as np
from .. import _image_size

_3d_spot_infos = ['height', 'z', 'x', 'y', 'background', 'sigma_z', 'sigma_x', 'sigma_y', 'sin_t', 'sin_p', 'eps']
_3d_infos = ['z', 'x', 'y']
_spot_coord_inds = [_3d_spot_infos.index(_info) for _info in _3d_infos]


class ImageCrop():
    """(ndim, 2) int32 box of [start, stop) limits (preprocess.py:17-104)."""

    def __init__(self, ndim, crop_array=None, single_im_size=_image_size):
        self.ndim = ndim
        self.array = np.zeros((ndim, 2), dtype=np.int32)
        if crop_array is None:
            self.array[:, 1] = np.array(single_im_size)
        else:
            self.update(crop_array)
        if len(single_im_size) == ndim:
            self.image_sizes = np.array(single_im_size, dtype=np.int32)

    def update(self, crop_array):
        _arr = np.array(crop_array, dtype=np.int32)
        if np.shape(_arr) == np.shape(self.array):
            self.array = _arr

    def to_slices(self):
        return tuple(slice(int(_s[0]), int(_s[1])) for _s in self.array)

    def inside(self, coords):
        _coords = np.array(coords)
        if _coords.ndim == 1:
            _coords = _coords[np.newaxis, :]
        elif _coords.ndim > 2:
            raise IndexError("Only support single or multiple coordinates")
        _mask = np.ones(len(_coords), dtype=bool)
        for _d in range(self.ndim):
            _mask &= (_coords[:, _d] >= self.array[_d, 0]) & (_coords[:, _d] <= self.array[_d, 1])
        return _mask

    def distance_to_edge(self, coord):
        _coord = np.array(coord)[:self.ndim]
        return np.min(np.abs(_coord[:, np.newaxis] - self.array))

    def crop_coords(self, coords):
        _coords = np.array(coords)
        return _coords[self.inside(coords)] - self.array[:, 0][np.newaxis, :]

    def overlap(self, crop2):
        _llim = np.max([self.array[:, 0], crop2.array[:, 0]], axis=0)
        _rlim = np.min([self.array[:, 1], crop2.array[:, 1]], axis=0)
        if (_llim > _rlim).any():
            return None
        return ImageCrop(len(_llim), np.array([_llim, _rlim]).transpose())

    def relative_overlap(self, crop2):
        _overlap = self.overlap(crop2)
        if _overlap is not None:
            _overlap.array = _overlap.array - self.array[:, 0][:, np.newaxis]
        return _overlap


class ImageCrop_3d(ImageCrop):
    """preprocess.py:106-137."""

    def __init__(self, crop_array=None, single_im_size=_image_size):
        super().__init__(3, crop_array, single_im_size)

    def crop_spots(self, spots_3d):
        _spots = spots_3d.copy()
        _mask = self.inside(_spots[:, 1:4])
        _cropped = _spots[_mask].copy()
        _cropped[:, 1:4] = np.array(_cropped[:, 1:4]) - self.array[:, 0][np.newaxis, :]
        return _cropped

    def overlap(self, crop2):
        _c = super().overlap(crop2)
        return None if _c is None else ImageCrop_3d(_c.array)

    def translate_drift(self, drift=None):
        _drift = np.zeros(self.ndim, dtype=np.int32) if drift is None else np.round(drift).astype(np.int32)
        _box = [[max(0, _l[0] - _d), min(_sz, _l[1] - _d)]
                for _l, _d, _sz in zip(self.array, _drift, self.image_sizes)]
        return ImageCrop_3d(np.array(_box, dtype=np.int32), self.image_sizes)


class Spots3D(np.ndarray):
    """ndarray view over an (N,11) spot table with bits/channels/pixel sizes (preprocess.py:139-251)."""

    def __new__(cls, input_array, bits=None, pixel_sizes=None, channels=None, copy_data=True,
                intensity_index=0, coordinate_indices=[1, 2, 3]):
        if copy_data:
            input_array = np.array(input_array).copy()
        if np.ndim(input_array) == 1:
            obj = np.asarray([input_array]).view(cls)
        elif np.ndim(input_array) == 2:
            obj = np.asarray(input_array).view(cls)
        else:
            raise IndexError('Spots3D class only creating 2D-array')
        if isinstance(bits, (int, np.integer)):
            obj.bits = np.ones(len(obj), dtype=np.int32) * int(bits)
        elif bits is not None and np.size(bits) == 1:
            obj.bits = np.ones(len(obj), dtype=np.int32) * int(np.ravel(bits)[0])
        elif bits is not None and len(bits) == len(obj):
            obj.bits = np.array(bits, dtype=np.int32)
        else:
            obj.bits = bits
        if isinstance(channels, bytes):
            channels = channels.decode()
        if isinstance(channels, (int, np.integer)):
            obj.channels = np.ones(len(obj), dtype=np.int32) * int(channels)
        elif channels is not None and isinstance(channels, str):
            obj.channels = np.array([channels] * len(obj))
        elif channels is not None and len(channels) == len(obj):
            obj.channels = np.array(channels)
        else:
            obj.channels = channels
        obj.pixel_sizes = np.array(pixel_sizes)
        obj.intensity_index = int(intensity_index)
        obj.coordinate_indices = np.array(coordinate_indices, dtype=np.int32)
        obj._3d_infos = _3d_infos
        obj._3d_spot_infos = _3d_spot_infos
        obj._spot_coord_inds = np.array(_spot_coord_inds)
        return obj

    def __getitem__(self, key):
        new_obj = super().__getitem__(key)
        for _name in ('bits', 'channels'):
            _v = getattr(self, _name, None)
            if _v is not None and np.ndim(_v) == 1 and isinstance(key, (slice, np.ndarray, int)):
                try:
                    setattr(new_obj, _name, _v[key])
                except (AttributeError, IndexError):
                    pass
        return new_obj

    def __array_finalize__(self, obj):
        if obj is None:
            return
        for _name in ('bits', 'channels', 'pixel_sizes', 'intensity_index', 'coordinate_indices',
                      '_3d_infos', '_3d_spot_infos', '_spot_coord_inds'):
            setattr(self, _name, getattr(obj, _name, None))

    def to_coords(self):
        _ci = getattr(self, 'coordinate_indices', None)
        if _ci is None:
            _ci = np.array([1, 2, 3])
        return np.array(self[:, _ci])

    def to_positions(self, pixel_sizes=None):
        _saved = getattr(self, 'pixel_sizes', None)
        if _saved is not None and np.ndim(_saved) > 0 and _saved.any():
            return self.to_coords() * np.array(_saved)[np.newaxis, :]
        if pixel_sizes is None:
            raise ValueError('pixel_sizes not given')
        return self.to_coords() * np.array(pixel_sizes)[np.newaxis, :]

    def to_intensities(self):
        return np.array(self[:, getattr(self, 'intensity_index', 0) or 0])
