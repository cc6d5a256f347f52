"""Row schema and light containers around the spot table (reference: classes/preprocess.py:13-251).

The 11-column row naming, ``ImageCrop`` / ``ImageCrop_3d`` ([start, stop) boxes) and the ``Spots3D`` ndarray view
(own implementations behind the reference's interface).  The reference's ``DaxProcesser`` step class is a CALLER of the
hot path and stays the reference's own: the copy the end-to-end parity fixtures run through lives with the
tests (tests/harness/dax_processer.py), not in this package.
"""
import numpy as np
from .. import _image_size

_3d_spot_infos = ['height', 'z', 'x', 'y', 'background', 'sigma_z', 'sigma_x', 'sigma_y', 'sin_t', 'sin_p', 'eps']
_3d_infos = ['z', 'x', 'y']
_spot_coord_inds = [_3d_spot_infos.index(_info) for _info in _3d_infos]


def _clip_box(lo, hi, size=None):
    """(ndim, 2) int32 [start, stop) table from two limit vectors, optionally clipped to [0, size]."""
    lo, hi = np.asarray(lo), np.asarray(hi)
    if size is not None:
        lo, hi = np.maximum(lo, 0), np.minimum(hi, np.asarray(size))
    return np.stack([lo, hi], axis=1).astype(np.int32)


class ImageCrop():
    """A box of [start, stop) limits per axis, ``.array`` (ndim, 2) int32 — interface of preprocess.py:17-104.

    ``inside`` keeps the reference's closed upper limit (a coordinate equal to ``stop`` counts as inside)."""

    def __init__(self, ndim, crop_array=None, single_im_size=_image_size):
        size = np.asarray(single_im_size)
        self.ndim = ndim
        self.array = np.zeros((ndim, 2), dtype=np.int32)
        if crop_array is None:
            self.array[:, 1] = size          # whole image
        else:
            self.update(crop_array)
        if size.size == ndim:
            self.image_sizes = size.astype(np.int32)

    def update(self, crop_array):
        box = np.array(crop_array, dtype=np.int32)
        if box.shape == self.array.shape:    # anything else is ignored, as in the reference
            self.array = box

    def to_slices(self):
        return tuple(slice(int(a), int(b)) for a, b in self.array)

    def inside(self, coords):
        pts = np.array(coords)
        if pts.ndim > 2:
            raise IndexError("Only support single or multiple coordinates")
        pts = np.atleast_2d(pts)[:, :self.ndim]
        return ((pts >= self.array[:, 0]) & (pts <= self.array[:, 1])).all(axis=1)

    def distance_to_edge(self, coord):
        pt = np.asarray(coord)[:self.ndim]
        return np.abs(pt[:, None] - self.array).min()

    def crop_coords(self, coords):
        pts = np.array(coords)
        return pts[self.inside(pts)] - self.array[:, 0]

    def _intersection(self, other):
        lo = np.maximum(self.array[:, 0], other.array[:, 0])
        hi = np.minimum(self.array[:, 1], other.array[:, 1])
        return None if (lo > hi).any() else _clip_box(lo, hi)

    def overlap(self, crop2):
        box = self._intersection(crop2)
        return None if box is None else ImageCrop(len(box), box)

    def relative_overlap(self, crop2):
        shared = self.overlap(crop2)
        if shared is not None:
            shared.array = shared.array - self.array[:, :1]
        return shared


class ImageCrop_3d(ImageCrop):
    """Three-axis box with the spot-table helpers of preprocess.py:106-137."""

    def __init__(self, crop_array=None, single_im_size=_image_size):
        ImageCrop.__init__(self, 3, crop_array, single_im_size)

    def crop_spots(self, spots_3d):
        zxy = slice(_spot_coord_inds[0], _spot_coord_inds[-1] + 1)
        kept = spots_3d[self.inside(spots_3d[:, zxy])].copy()
        kept[:, zxy] = np.array(kept[:, zxy]) - self.array[:, 0]
        return kept

    def overlap(self, crop2):
        box = self._intersection(crop2)
        return None if box is None else ImageCrop_3d(box)

    def translate_drift(self, drift=None):
        shift = 0 if drift is None else np.round(drift).astype(np.int32)
        return ImageCrop_3d(_clip_box(self.array[:, 0] - shift, self.array[:, 1] - shift, self.image_sizes),
                            self.image_sizes)


def _row_labels(value, n, is_scalar):
    """Per-row label vector for ``Spots3D.bits`` / ``.channels``: a scalar is repeated, a length-n sequence is kept,
    anything else is stored as given."""
    if value is None:
        return None
    if is_scalar(value):
        return np.repeat(value, n)
    if np.size(value) == n and np.ndim(value) == 1:
        return np.array(value)
    return value


class Spots3D(np.ndarray):
    """(N, 11) spot table as an ndarray view carrying per-row ``bits`` / ``channels`` and the pixel size
    (interface of preprocess.py:139-251): indexing rows indexes the labels too."""

    _meta = ('bits', 'channels', 'pixel_sizes', 'intensity_index', 'coordinate_indices',
             '_3d_infos', '_3d_spot_infos', '_spot_coord_inds')

    def __new__(cls, input_array, bits=None, pixel_sizes=None, channels=None, copy_data=True,
                intensity_index=0, coordinate_indices=[1, 2, 3]):
        table = np.array(input_array) if copy_data else np.asarray(input_array)
        if table.ndim not in (1, 2):
            raise IndexError('Spots3D class only creating 2D-array')
        obj = np.atleast_2d(table).view(cls)
        n = len(obj)
        bits = _row_labels(bits, n, lambda v: np.size(v) == 1)
        if bits is not None and np.ndim(bits) == 1 and len(bits) == n:
            bits = np.asarray(bits).astype(np.int32)
        obj.bits = bits
        if isinstance(channels, bytes):
            channels = channels.decode()
        channels = _row_labels(channels, n, lambda v: isinstance(v, (str, int, np.integer)))
        if isinstance(channels, np.ndarray) and channels.dtype.kind in 'iu':
            channels = channels.astype(np.int32)
        obj.channels = channels
        obj.pixel_sizes = np.array(pixel_sizes)
        obj.intensity_index = int(intensity_index)
        obj.coordinate_indices = np.array(coordinate_indices, dtype=np.int32)
        obj._3d_infos, obj._3d_spot_infos = _3d_infos, _3d_spot_infos
        obj._spot_coord_inds = np.array(_spot_coord_inds)
        return obj

    def __array_finalize__(self, obj):
        if obj is not None:
            for name in self._meta:
                setattr(self, name, getattr(obj, name, None))

    def __getitem__(self, key):
        out = np.ndarray.__getitem__(self, key)
        if isinstance(key, (slice, np.ndarray, int)):      # a row selection: carry the matching labels along
            for name in ('bits', 'channels'):
                labels = getattr(self, name, None)
                if labels is not None and np.ndim(labels) == 1:
                    try:
                        setattr(out, name, labels[key])
                    except (AttributeError, IndexError):
                        pass
        return out

    def to_coords(self):
        cols = getattr(self, 'coordinate_indices', None)
        return np.array(self[:, np.array([1, 2, 3]) if cols is None else cols])

    def to_positions(self, pixel_sizes=None):
        own = getattr(self, 'pixel_sizes', None)
        if own is not None and np.ndim(own) > 0 and own.any():
            pixel_sizes = own
        elif pixel_sizes is None:
            raise ValueError('pixel_sizes not given')
        return self.to_coords() * np.asarray(pixel_sizes)[None, :]

    def to_intensities(self):
        return np.array(self[:, getattr(self, 'intensity_index', 0) or 0])
