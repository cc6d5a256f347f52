"""Drop-in for the two functions of the reference's ``corrections.py`` that sit on the per-FOV path
(``io_tools/load.py:330,342`` call them): ``Remove_Hot_Pixels`` (:490-510) and ``Z_Shift_Correction``
(:479-487).  Everything else in that module is offline calibration (out of scope)."""
import ctypes as C
import numpy as np

from . import _lib as L
from .correction_tools.filter import Remove_Hot_Pixels  # same arithmetic as corrections.py:490-510  # noqa: F401


def Z_Shift_Correction(im, dtype=np.uint16, normalization=False, verbose=False):
    """corrections.py:479-487 — ``im / median_z[:,None,None] * median(im)`` cast to ``dtype``.
    (Both branches of ``normalization`` are identical in the reference.)  The float32 arithmetic of the
    production call ``Z_Shift_Correction(im.astype(np.float32), dtype=np.uint16)`` (io_tools/load.py:342)
    runs on the device; medians by radix select."""
    if verbose:
        print("-- correcting Z axis illumination shifts.")
    a = L.as_stack_array(im)
    if np.dtype(dtype) != np.uint16:
        raise NotImplementedError("device Z_Shift_Correction produces uint16 (the pipeline's output_dtype)")
    out = np.empty(a.shape, dtype=np.uint16)
    L.check(L.lib().ia3_z_shift_correction(L.ptr(a), L.dtype_code(a), a.shape[0], a.shape[1], a.shape[2],
                                           L.ptr(out), None))
    return out
