"""Drop-in for the two seeding functions of the reference's ``visual_tools.py`` that the legacy per-cell
fitter (``classes/__init__.py:57-88 _fit_single_image``) calls: ``get_seed_points_base`` (:348-381) and
``get_seed_in_distance`` (:1775-1870).  Filters, rank tests and compaction run on the device
(``ia3_seed_in_distance``: seed.hip); the percentile threshold (``seed_by_per``) is evaluated here.

The reference functions use ``np.float`` / ``np.int`` (removed in NumPy 1.24); the semantics restated
here are those of ``float`` / ``int`` which those aliases were.  Behaviours kept as written there:
* without ``dynamic`` (or without ``center``) the base function is called WITHOUT
  ``background_gfilt_size`` (its default 10 applies) and, with a ``center``, the seeds stay in crop
  coordinates and are not distance-filtered (:1851-1858);
* heights are int64 differences of the truncated rank-filter outputs; threshold test is a strict ``>``.
"""
import ctypes as C
import numpy as np

from . import _lib as L


def _prep(im):
    if not isinstance(im, np.ndarray):
        raise TypeError("im should be a numpy.ndarray")
    if im.ndim != 3:
        raise IndexError("im should be a 3-D stack")
    if im.dtype == np.uint16 or im.dtype == np.float32:
        return np.ascontiguousarray(im)
    if im.dtype.kind in "ui" and im.size and im.min() >= 0 and im.max() <= 65535:
        return np.ascontiguousarray(im, dtype=np.uint16)
    raise NotImplementedError("legacy seeding supports uint16 and float32 stacks (got %s)" % im.dtype)


def _call(im, center, num_seeds, seed_radius, gfilt_size, background_gfilt_size, filt_size, th_seed, dynamic,
          dynamic_iters, min_dynamic_seeds, hot_pix_th):
    im = _prep(im)
    p = L.LegacySeedParams()
    p.num_seeds = int(num_seeds)
    p.seed_radius = float(seed_radius)
    p.gfilt_size = float(gfilt_size or 0)
    p.background_gfilt_size = float(background_gfilt_size or 0)
    p.filt_size = int(filt_size)
    p.th_seed = float(th_seed)
    p.dynamic = int(bool(dynamic))
    p.dynamic_iters = int(dynamic_iters)
    p.min_dynamic_seeds = int(min_dynamic_seeds)
    p.hot_pix_th = int(hot_pix_th)
    cptr = None
    if center is not None:
        c = np.ascontiguousarray(center, dtype=np.float64)
        cptr = L.dptr(c)
    cap = 4096
    n = C.c_int(0)
    while True:
        out = np.empty((cap, 4), dtype=np.int64)
        rc = L.lib().ia3_seed_in_distance(L.ptr(im), L.dtype_code(im), im.shape[0], im.shape[1], im.shape[2],
                                          cptr, C.byref(p), L.ptr(out), cap, C.byref(n))
        if rc == L.IA3_ECAPACITY and n.value > cap:
            cap = n.value
            continue
        L.check(rc)
        return out[:n.value]


def get_seed_points_base(im, gfilt_size=0.75, background_gfilt_size=10, filt_size=3,
                         th_seed=300, hot_pix_th=0, return_h=False):
    """visual_tools.py:348-381 — returns a (3,N) or (4,N) int64 array [z,x,y(,h)] in np.where order."""
    im = _prep(im)
    # the whole-image branch of the device entry is the base function followed by the height sort; undo the
    # sort to give np.where order back
    seeds = _call(im, None, 0, 0., gfilt_size, 10, filt_size, th_seed, False, 1, 0, hot_pix_th) \
        if background_gfilt_size == 10 else _base_with_background(im, gfilt_size, background_gfilt_size, filt_size,
                                                                 th_seed, hot_pix_th)
    order = np.lexsort((seeds[:, 2], seeds[:, 1], seeds[:, 0])) if len(seeds) else np.zeros(0, dtype=int)
    seeds = seeds[order]
    return seeds.T.copy() if return_h else seeds[:, :3].T.copy()


def _base_with_background(im, gfilt_size, background_gfilt_size, filt_size, th_seed, hot_pix_th):
    # dynamic branch with a single level, a centre in the middle and a radius that covers the stack: the base
    # function with an explicit background sigma
    Z, X, Y = im.shape
    r = 4. * float(max(Z, X, Y)) + 8.
    return _call(im, (Z / 2., X / 2., Y / 2.), 0, r, gfilt_size, background_gfilt_size, filt_size, th_seed,
                 True, 1, 0, hot_pix_th)


def get_seed_in_distance(im, center=None, num_seeds=0, seed_radius=30,
                         gfilt_size=0.75, background_gfilt_size=10, filt_size=3,
                         seed_by_per=False, th_seed_percentile=95,
                         th_seed=300,
                         dynamic=True, dynamic_iters=10, min_dynamic_seeds=2,
                         distance_to_edge=1, hot_pix_th=4,
                         return_h=False, verbose=False):
    """visual_tools.py:1775-1870 — seeds within ``seed_radius`` of ``center`` (z,x,y), brightest first.
    Returns an (N,3) (or (N,4) with heights) int64 array."""
    if center is not None and len(center) != 3:
        raise ValueError('wrong input dimension of center!')
    if seed_by_per:
        from scipy.stats import scoreatpercentile
        _im_ints = im[np.isnan(im) == False].astype(float)
        _th_seed = scoreatpercentile(_im_ints, th_seed_percentile) - \
            scoreatpercentile(_im_ints, 100 - th_seed_percentile)
    else:
        _th_seed = th_seed
    if verbose:
        print(f"-- seeding with threshold: {_th_seed}, per={th_seed_percentile}")
    if center is not None:
        # with a centre the reference uses th_seed (not the percentile threshold) in both branches (:1834,:1853)
        seeds = _call(im, center, num_seeds, seed_radius, gfilt_size, background_gfilt_size, filt_size, th_seed,
                      dynamic, dynamic_iters, min_dynamic_seeds, hot_pix_th)
    else:
        seeds = _call(im, None, num_seeds, seed_radius, gfilt_size, 10, filt_size, _th_seed,
                      False, 1, 0, hot_pix_th)
    return seeds.copy() if return_h else seeds[:, :3].copy()
