"""Drop-in for the reference's ``correction_tools/filter.py`` (HIP kernels: gauss.hip, hotpix.hip)."""
import ctypes as C
import numpy as np

from .. import _lib as L


def gaussian_filter(image, sigma, mode='reflect', truncate=4.0):
    """scipy.ndimage.gaussian_filter twin for uint16/float32 3-D stacks (modes reflect / nearest),
    bit-identical to SciPy: float64 accumulation in NI_Correlate1D order, per-axis re-quantisation."""
    a = L.as_stack_array(image)
    w, r = L.gaussian_taps(sigma, truncate)
    out = np.empty_like(a)
    m = {'reflect': L.MODE_REFLECT, 'nearest': L.MODE_NEAREST}[mode]
    L.check(L.lib().ia3_gaussian_filter(L.ptr(a), L.dtype_code(a), a.shape[0], a.shape[1], a.shape[2],
                                        C.c_double(sigma), C.c_double(truncate), m, L.dptr(w), r, L.ptr(out)))
    return out


def gaussian_deconvolution(im, gfilt_size=2, niter=1):
    """correction_tools/filter.py:4-11 — image divided by its Gaussian blur, ``niter`` times."""
    decon_im = im.copy().astype(np.float32)
    for _iter in np.arange(niter):
        decon_im = decon_im / gaussian_filter(decon_im, gfilt_size)
    return decon_im


def gaussian_high_pass_filter(image, sigma=5, truncate=2):
    """correction_tools/filter.py:14-19 — image - lowpass(mode nearest), clamped at 0, input dtype."""
    a = L.as_stack_array(image)
    w, r = L.gaussian_taps(sigma, truncate)
    out = np.empty_like(a)
    L.check(L.lib().ia3_gaussian_highpass(L.ptr(a), L.dtype_code(a), a.shape[0], a.shape[1], a.shape[2],
                                          C.c_double(sigma), C.c_double(truncate), L.dptr(w), r, L.ptr(out)))
    return out


def Remove_Hot_Pixels(im, dtype=np.uint16, hot_pix_th=0.50, hot_th=4,
                      interpolation_style='nearest', verbose=False):
    """correction_tools/filter.py:22-42 (same arithmetic as corrections.py:490-510)."""
    if verbose:
        print("-- removing hot pixels")
    a = L.as_stack_array(im)
    out = np.empty_like(a)
    n_hot = C.c_int(0)
    L.check(L.lib().ia3_remove_hot_pixels(L.ptr(a), L.dtype_code(a), a.shape[0], a.shape[1], a.shape[2],
                                          C.c_double(hot_pix_th), C.c_double(hot_th), L.ptr(out), C.byref(n_hot)))
    if n_hot.value == 0:
        return im                                                            # :32-33
    if interpolation_style != 'nearest':
        return im.copy().astype(dtype)   # the reference only implements 'nearest' (:37)
    return out.astype(dtype)
