"""Drop-in for the hot-path part of the reference's ``alignment_tools.py`` (:278-353):
pixel-level drift from max-projections by FFT cross-correlation, on the device (fft_align.hip)."""
import ctypes as C
import numpy as np

from . import _lib as L


def blurnorm2d(im, gb):
    """alignment_tools.py:278-283 needs ``cv2.blur``; OpenCV is an optional dependency of the reference
    used only when ``gb > 1`` (off the default path, ``fft_filt_size=0`` at correction_tools/alignment.py:141)."""
    import cv2  # noqa: F401  (raises ImportError where the reference would)
    im_ = im.astype(np.float32)
    return im_ / cv2.blur(im_, (gb, gb))


def fftalign_2d(im1, im2, center=[0, 0], max_disp=150, plt_val=False):
    """alignment_tools.py:286-328 — (xt, yt) of the windowed peak of the normalised cross-correlation."""
    a = np.ascontiguousarray(im1, dtype=np.float64)
    b = np.ascontiguousarray(im2, dtype=np.float64)
    if a.ndim != 2 or b.ndim != 2:
        raise IndexError("fftalign_2d takes 2-D images")
    c = np.ascontiguousarray(center, dtype=np.float64)
    out = (C.c_int * 2)()
    L.check(L.lib().ia3_fftalign_2d(L.dptr(a), a.shape[0], a.shape[1], L.dptr(b), b.shape[0], b.shape[1],
                                    L.dptr(c), C.c_double(float(max_disp)), out))
    return int(out[0]), int(out[1])


def fft3d_from2d(im1, im2, gb=5, max_disp=150):
    """alignment_tools.py:330-353 — integer [tz, tx, ty]; max-projections and FFTs on the device.
    ``im1``/``im2`` may be ndarrays or DeviceStacks."""
    if gb > 1:
        # cv2-normalised variant: projections on the host exactly as the reference, FFT on the device
        i1, i2 = _host(im1), _host(im2)
        im1_ = blurnorm2d(np.max(i1, 0), gb)
        im2_ = blurnorm2d(np.max(i2, 0), gb)
        tx, ty = fftalign_2d(im1_, im2_, center=[0, 0], max_disp=max_disp)
        sx, sy = im1_.shape
        im1_t = blurnorm2d(np.max(i1[:, max(tx, 0):sx + tx, max(ty, 0):sy + ty], axis=-1), gb)
        im2_t = blurnorm2d(np.max(i2[:, max(-tx, 0):sx - tx, max(-ty, 0):sy - ty], axis=-1), gb)
        tz, _ = fftalign_2d(im1_t, im2_t, center=[0, 0], max_disp=max_disp)
        return np.array([tz, tx, ty])
    out = (C.c_int * 3)()
    if isinstance(im1, L.DeviceStack) and isinstance(im2, L.DeviceStack):
        L.check(L.lib().ia3_fft3d_from2d_dev(im1._h, im2._h, C.c_double(float(max_disp)), out))
    else:
        a, b = L.as_stack_array(_host(im1)), L.as_stack_array(_host(im2))
        if a.shape != b.shape or a.dtype != b.dtype:
            raise IndexError("fft3d_from2d needs two stacks of the same shape and dtype")
        L.check(L.lib().ia3_fft3d_from2d(L.ptr(a), L.ptr(b), L.dtype_code(a), a.shape[0], a.shape[1], a.shape[2],
                                         C.c_double(float(max_disp)), out))
    return np.array([out[0], out[1], out[2]])


def _host(im):
    return im.download() if isinstance(im, L.DeviceStack) else im
