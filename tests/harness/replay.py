"""TEST HARNESS — flat replay of the operator sequences behind the ``daxp`` / ``seg`` fixtures.

The fixtures were produced by running the reference's step-by-step driver on a synthetic .dax movie
(oracle/make_golden.py::daxp_golden, seg_golden).  What they pin is a SEQUENCE OF OPERATOR CALLS with that driver's
arithmetic — image-dtype hot-pixel votes, float64 bleedthrough accumulation with min-max rescale, rescaled illumination,
drift applied before the chromatic field, fits on label bounding boxes — not a class.  Each function below is one such
call on resident stacks (``{channel: DeviceStack}``), straight through the C ABI.
"""
import ctypes as C
import numpy as np
from imageanalysis3_amd import _lib as L
from imageanalysis3_amd.io_tools.load import load_dax_resident, split_im_by_channels, DeviceBuffer


def _buf(p):
    return p if isinstance(p, DeviceBuffer) else DeviceBuffer(p)


def _swap(stacks, ch, new):
    stacks[ch].free()
    stacks[ch] = new


def load_channels(dax, channels, all_channels, im_size, n_buffer=0, n_empty=0):
    """Raw movie -> one resident (Z, X, Y) stack per channel."""
    raw = load_dax_resident(dax)
    try:
        ims = split_im_by_channels(raw, channels, all_channels=all_channels, single_im_size=np.array(im_size, dtype=int),
                                   num_buffer_frames=n_buffer, num_empty_frames=n_empty)
    finally:
        raw.free()
    return dict(zip(channels, ims))


def hot_pixels_in_image_dtype(stacks, channels, hot_pixel_th=0.5, hot_pixel_num_th=4):
    """correction_tools.filter.Remove_Hot_Pixels on the uint16 stack itself (sums wrap), in place."""
    for ch in channels:
        n = C.c_int(0)
        L.check(L.lib().ia3_remove_hot_pixels_dev(stacks[ch]._h, C.c_double(hot_pixel_th), C.c_double(hot_pixel_num_th), 0,
                                                  C.byref(n)))


def bleedthrough_rescaled(stacks, channels, profile, rescale):
    """Channel mix accumulated in float64, optional min-max rescale to the uint16 range, clip, truncate."""
    pf = _buf(profile)
    n = len(channels)
    outs = [L.DeviceStack.empty(stacks[ch].shape, np.uint16) for ch in channels]
    a_in = (C.c_void_p * n)(*[stacks[ch]._h for ch in channels])
    a_out = (C.c_void_p * n)(*[o._h for o in outs])
    L.check(L.lib().ia3_bleedthrough_rescale_dev(a_in, n, pf.ptr, pf.dtype_code, int(bool(rescale)), a_out))
    for ch, o in zip(channels, outs):
        _swap(stacks, ch, o)


def illumination_rescaled(stacks, channels, profiles, rescale):
    for ch in channels:
        pf = _buf(profiles[ch])
        out = L.DeviceStack.empty(stacks[ch].shape, np.uint16)
        L.check(L.lib().ia3_illumination_rescale_dev(stacks[ch]._h, pf.ptr, pf.dtype_code, int(bool(rescale)), out._h))
        _swap(stacks, ch, out)


def warp_drift_then_field(stacks, channels, drift, fields):
    """Cubic resampling at (grid - drift) + field (the step driver's order; field_dtype + 16 in the C ABI); ``fields[ch]``
    None or absent = drift only."""
    d = np.ascontiguousarray(drift, dtype=np.float64)
    for ch in channels:
        f = fields.get(ch)
        fb = None if f is None else _buf(f)
        out = L.DeviceStack.empty(stacks[ch].shape, stacks[ch].dtype)
        L.check(L.lib().ia3_warp3d_dev(stacks[ch]._h, L.dptr(d), None if fb is None else fb.ptr,
                                       16 if fb is None else fb.dtype_code + 16, 3, L.MODE_NEAREST, C.c_double(0.0), out._h))
        _swap(stacks, ch, out)


def highpass(stacks, channels, sigma=3, truncate=2):
    w, r = L.gaussian_taps(sigma, truncate)
    for ch in channels:
        out = L.DeviceStack.empty(stacks[ch].shape, stacks[ch].dtype)
        L.check(L.lib().ia3_gaussian_highpass_dev(stacks[ch]._h, C.c_double(sigma), C.c_double(truncate), L.dptr(w), int(r), out._h))
        _swap(stacks, ch, out)


def label_box(mask, margin=1):
    """[start, stop) per axis around the non-zero voxels of ``mask``, grown by ``margin`` and clipped to the image."""
    hit = np.nonzero(np.asarray(mask))
    return np.array([[max(int(ix.min()) - margin, 0), min(int(ix.max()) + 1 + margin, n)] for ix, n in zip(hit, np.shape(mask))])


def labels_around(label_image, centres_zxy, radius):
    """Most frequent positive label in the (2 radius + 1)^3 cube around every rounded centre (indices clamped to the
    image; ties: smallest label), -1 where the cube holds none."""
    lab = np.asarray(label_image)
    c = np.round(np.asarray(centres_zxy, dtype=np.float64)).astype(np.int32)
    r = np.arange(-radius, radius + 1)
    off = np.stack(np.meshgrid(r, r, r, indexing="ij"), axis=-1).reshape(-1, 3)
    out = np.full(len(c), -1, dtype=np.int32)
    for k in range(len(c)):
        idx = np.clip(c[k] + off, 0, np.array(lab.shape) - 1)
        cube = lab[idx[:, 0], idx[:, 1], idx[:, 2]]
        vals, counts = np.unique(cube[cube > 0], return_counts=True)
        if len(vals):
            out[k] = vals[np.argmax(counts)]
    return out


def fit_in_labels(stack, channel, label_image, drift, th_seed=500, num_spots=None, search_radius=3):
    """Per positive label: fit_fov_image on the label's bounding box (margin one voxel, shifted by the drift, cut on the
    device), rows moved back to image coordinates, kept when their neighbourhood votes for that label."""
    from imageanalysis3_amd.spot_tools.fitting import fit_fov_image
    labels = np.unique(label_image)
    tables, owners = [], []
    for lab in labels[labels > 0]:
        mask = label_image == lab
        box = label_box(mask, 1)
        # the box moves against the drift by whole voxels and is clipped to the image (preprocess.py:95-98)
        shift = np.round(np.asarray(drift, dtype=np.float64)).astype(np.int32)
        lo = np.maximum(box[:, 0] - shift, 0).astype(int)
        hi = np.minimum(box[:, 1] - shift, np.array(mask.shape)).astype(int)
        crop = stack.crop(np.stack([lo, hi], axis=1))
        try:
            rows = fit_fov_image(crop, str(channel), th_seed=th_seed, max_num_seeds=num_spots, verbose=False)
        finally:
            crop.free()
        if len(rows) == 0:
            continue
        rows = np.array(rows)
        rows[:, 1:4] = rows[:, 1:4] + lo
        keep = labels_around(mask, rows[:, 1:4], search_radius) > 0
        if keep.any():
            tables.append(rows[keep])
            owners.append(np.full(int(keep.sum()), lab, dtype=np.int32))
    if not tables:
        return np.array([]), np.array([])
    return np.concatenate(tables), np.concatenate(owners)


def free_all(stacks):
    for s in stacks.values():
        s.free()
    stacks.clear()
