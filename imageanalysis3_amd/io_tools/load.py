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


def find_image_background(im, dtype=_image_dtype, bin_size=10, make_plot=False, max_iter=10):
    """Histogram-peak background level (io_tools/load.py:642-687)."""
    import scipy.signal
    if dtype is None:
        dtype = im.dtype
    _cts, _bins = np.histogram(im, bins=np.arange(np.iinfo(dtype).min, np.iinfo(dtype).max, bin_size))
    _peaks = []
    _height = np.size(im) / 50
    _iter = 0
    while len(_peaks) == 0:
        _height = _height / 2
        _peaks, _params = scipy.signal.find_peaks(_cts, height=_height)
        _iter += 1
        if _iter > max_iter:
            break
    if _iter > max_iter:
        _background = np.nanmedian(im)
    else:
        _sel_peak = _peaks[np.argmax(_params['peak_heights'])]
        _background = (_bins[_sel_peak] + _bins[_sel_peak + 1]) / 2
    return _background
