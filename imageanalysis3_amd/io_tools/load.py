"""reference: io_tools/load.py — only ``find_image_background`` (:642-687) so far.

Host-side NumPy (a few thousand voxels per call).  SURVEY.md §8(f4) lists the batched
histogram-mode kernel for ``normalize_local`` as a "next" row; until it exists this helper is
the reference's own arithmetic on the host, outside the measured hot path.
"""
import ctypes as C
import numpy as np
from .. import _image_dtype
from .. import _lib as L


def illumination_correction(im, profile, output_dtype=np.uint16):
    """io_tools/load.py:373-384 — ``(im.astype(np.float32) / profile[np.newaxis,:]).astype(output_dtype)``
    on the device.  ``im`` (Z,X,Y) uint16, ``profile`` (X,Y) float32 or float64."""
    a = np.ascontiguousarray(im)
    if a.dtype != np.uint16 or np.dtype(output_dtype) != np.uint16:
        raise TypeError("illumination_correction takes and returns uint16 stacks")
    p = np.ascontiguousarray(profile)
    if p.dtype not in (np.float32, np.float64):
        p = p.astype(np.float64)
    if p.shape != a.shape[1:]:
        raise IndexError(f"illumination profile shape {p.shape} should be {a.shape[1:]}")
    out = np.empty_like(a)
    L.check(L.lib().ia3_illumination_correct(L.ptr(a), a.shape[0], a.shape[1], a.shape[2], L.ptr(p),
                                             1 if p.dtype == np.float32 else 2, L.ptr(out)))
    return out


def bleedthrough_correction(ims, bleed_profile, output_dtype=np.uint16):
    """io_tools/load.py:348-370 — ``new_i = sum_j ims[j] * bleed_profile[i, j]`` clipped to uint16, on the
    device.  ``ims``: list of C (Z,X,Y) uint16 stacks (the corr_channels), ``bleed_profile`` (C,C,X,Y)."""
    ims = [np.ascontiguousarray(_im) for _im in ims]
    n_ch = len(ims)
    if any(_im.dtype != np.uint16 for _im in ims) or np.dtype(output_dtype) != np.uint16:
        raise TypeError("bleedthrough_correction takes and returns uint16 stacks")
    p = np.ascontiguousarray(bleed_profile)
    if p.dtype not in (np.float32, np.float64):
        p = p.astype(np.float64)
    if p.shape != (n_ch, n_ch) + ims[0].shape[1:]:
        raise IndexError(f"bleed_profile shape {p.shape} should be {(n_ch, n_ch) + ims[0].shape[1:]}")
    outs = [np.empty_like(_im) for _im in ims]
    arr_in = (C.c_void_p * n_ch)(*[_im.ctypes.data for _im in ims])
    arr_out = (C.c_void_p * n_ch)(*[_o.ctypes.data for _o in outs])
    Z, X, Y = ims[0].shape
    L.check(L.lib().ia3_bleedthrough_correct(arr_in, n_ch, Z, X, Y, L.ptr(p), 1 if p.dtype == np.float32 else 2,
                                             arr_out))
    return outs


def _background_edges(dtype, bin_size):
    """The reference's bin edges (io_tools/load.py:655-658) as the float64 vector the device entry takes."""
    info = np.iinfo(dtype)
    return np.ascontiguousarray(np.arange(info.min, info.max, bin_size), dtype=np.float64)


def _stack3(im):
    if not isinstance(im, np.ndarray):
        raise TypeError("im should be a numpy.ndarray")
    a = im if im.ndim == 3 else im.reshape((1,) * (3 - im.ndim) + im.shape) if im.ndim < 3 else im.reshape(-1, *im.shape[-2:])
    return L.as_stack_array(a)


def find_image_background(im, dtype=_image_dtype, bin_size=10, make_plot=False, max_iter=10):
    """Histogram-peak background level (io_tools/load.py:642-687), computed by ``ia3_find_background(_dev)``
    (background.hip).  ``im``: uint16/float32 ndarray or a resident ``DeviceStack``.  ``make_plot`` is accepted
    for signature compatibility and ignored (plotting is out of scope)."""
    if dtype is None:
        dtype = im.dtype
    edges = _background_edges(dtype, bin_size)
    out = C.c_double(0.0)
    if isinstance(im, L.DeviceStack):
        L.check(L.lib().ia3_find_background_dev(im._h, L.dptr(edges), len(edges), int(max_iter), C.byref(out)))
    else:
        a = _stack3(im)
        L.check(L.lib().ia3_find_background(L.ptr(a), L.dtype_code(a), a.shape[0], a.shape[1], a.shape[2],
                                            L.dptr(edges), len(edges), int(max_iter), C.byref(out)))
    return np.float64(out.value)


def find_local_backgrounds(stack, centers_zxy, crop_size, dtype=_image_dtype, bin_size=10, make_plot=False,
                           max_iter=10):
    """``[find_image_background(im[generate_neighboring_crop(c, crop_size).to_slices()], ...) for c in centers]``
    (spot_tools/fitting.py:249-256) in one launch on a resident stack."""
    if dtype is None:
        dtype = stack.dtype
    edges = _background_edges(dtype, bin_size)
    c = np.ascontiguousarray(centers_zxy, dtype=np.float32).reshape(-1, 3)
    out = np.empty(len(c), dtype=np.float64)
    L.check(L.lib().ia3_local_background_dev(stack._h, L.ptr(c), len(c), int(crop_size), L.dptr(edges), len(edges),
                                             int(max_iter), L.dptr(out)))
    return out
