"""Drop-in for the reference's ``correction_tools/translate.py`` (warp.hip)."""
import ctypes as C
import time
import numpy as np

from .. import _lib as L

_MODES = {'constant': L.MODE_CONSTANT, 'nearest': L.MODE_NEAREST}


def warp_3d_image(image, drift, chromatic_profile=None,
                  warp_order=1, border_mode='constant',
                  verbose=False):
    """correction_tools/translate.py:5-31 — resample ``image`` at ``grid (+ chromatic_profile) - drift``.

    Same call covers the production twins (io_tools/load.py:438-453, classes/preprocess.py:918-946) with
    ``warp_order=3, border_mode='nearest'``.  ``cval`` is ``np.min(image)`` as in the reference (:29).
    The (3,Z,X,Y) float64 coordinate grid the reference builds is never materialised."""
    _start_time = time.time()
    a = L.as_stack_array(image)
    if border_mode not in _MODES:
        raise NotImplementedError(f"border_mode {border_mode!r}: 'constant' and 'nearest' are implemented")
    _drift = np.ascontiguousarray(np.array(drift, dtype=np.float64).reshape(-1)[:3])
    if len(_drift) != 3:
        raise IndexError("drift should have 3 components (z,x,y)")
    field, fdt = None, 0
    if chromatic_profile is not None:
        cp = np.asarray(chromatic_profile)
        if cp.shape != (3,) + a.shape:
            raise IndexError(f"chromatic_profile shape {cp.shape} should be {(3,) + a.shape}")
        if cp.dtype == np.float32:
            field, fdt = np.ascontiguousarray(cp), 1
        else:
            field, fdt = np.ascontiguousarray(cp, dtype=np.float64), 2
    out = np.empty_like(a)
    L.check(L.lib().ia3_warp3d(L.ptr(a), L.dtype_code(a), a.shape[0], a.shape[1], a.shape[2], L.dptr(_drift),
                               L.ptr(field) if field is not None else None, fdt, int(warp_order),
                               _MODES[border_mode], C.c_double(float(np.min(a))), L.ptr(out)))
    if verbose:
        print(f"-- finish warp image in {time.time()-_start_time:.3f}s. ")
    return out
