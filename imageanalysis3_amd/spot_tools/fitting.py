"""Drop-in for the reference's ``spot_tools/fitting.py`` — same names, signatures and error
behaviour, computed by the HIP kernels of libia3.so (no CPU fallback).

All ``file:line`` citations are relative to the reference tree.
"""
import ctypes as C
import time
import numpy as np

from .. import _lib as L
from ..External import Fitting_v4


def get_seeds(im, max_num_seeds=None, th_seed=150,
              th_seed_per=95, use_percentile=False,
              sel_center=None, seed_radius=30,
              gfilt_size=0.75, background_gfilt_size=7.5,
              filt_size=3, min_edge_distance=2,
              use_dynamic_th=True, dynamic_niters=10, min_dynamic_seeds=1,
              remove_hot_pixel=True, hot_pixel_th=3,
              return_h=False, verbose=False,
              ):
    """spot_tools/fitting.py:20-154 — DoG local-maximum seeding.

    Returns (N,3) float64 [z,x,y] (or (N,4) with the DoG height), brightest first.
    Device path: ia3_dog_seed (gauss.hip + seed.hip)."""
    if not isinstance(im, np.ndarray):
        raise TypeError(f"image given should be a numpy.ndarray, but {type(im)} is given.")
    if th_seed_per >= 100 or th_seed_per <= 50:
        use_percentile = False
        print(f"th_seed_per should be a percentile > 50, invalid value given ({th_seed_per}), so not use percentile here.")
    if sel_center is not None:                                               # :56-68
        if len(sel_center) != len(np.shape(im)):
            raise IndexError("num of dimensions should match for selected center and image given.")
        _center = np.array(sel_center, dtype=int)
        _llims = np.max([np.zeros(len(im.shape)), _center - seed_radius], axis=0)
        _rlims = np.min([np.array(im.shape), _center + seed_radius], axis=0)
        _lims = np.array(np.transpose(np.stack([_llims, _rlims])), dtype=int)
        _im = im[tuple(slice(_l, _r) for _l, _r in _lims)]
        _local_edges = _llims
    else:
        _local_edges = np.zeros(len(np.shape(im)))
        _im = im
    if use_percentile:                                                       # :75-76 (whole image)
        _th_seed = _score_at_percentile(im, th_seed_per) - _score_at_percentile(im, (100 - th_seed_per) / 2)
    else:
        _th_seed = th_seed
    if verbose:
        _start_time = time.time()
        print(f"-- start seeding image, th={_th_seed:.2f}", end='')
    _a = L.as_stack_array(_im)
    _p, _keep = L.make_seed_params(_th_seed, gfilt_size=gfilt_size, background_gfilt_size=background_gfilt_size,
                                   filt_size=filt_size, min_edge_distance=min_edge_distance,
                                   use_dynamic_th=use_dynamic_th, dynamic_niters=dynamic_niters,
                                   min_dynamic_seeds=min_dynamic_seeds, remove_hot_pixel=remove_hot_pixel,
                                   hot_pixel_th=hot_pixel_th, max_num_seeds=max_num_seeds)
    _cap = 16384
    while True:
        _out = np.empty((_cap, 4), dtype=np.float64)
        _n, _th_used = C.c_int(0), C.c_double(0)
        _rc = L.lib().ia3_dog_seed(L.ptr(_a), L.dtype_code(_a), _a.shape[0], _a.shape[1], _a.shape[2],
                                   C.byref(_p), L.dptr(_out), _cap, C.byref(_n), C.byref(_th_used))
        if _rc == L.IA3_ECAPACITY and _n.value > _cap:
            _cap = _n.value
            continue
        L.check(_rc)
        break
    _final = _out[:_n.value].copy()
    _final[:, :3] += _local_edges[np.newaxis, :]
    if not return_h:
        _final = _final[:, :3].copy()
    if verbose:
        print(f"->{_th_used.value:.2f}, found {len(_final)} seeds in {time.time()-_start_time:.2f}s")
    return _final


def _score_at_percentile(a, per):
    """``scipy.stats.scoreatpercentile(a, per)`` (what spot_tools/fitting.py:76 calls) with its arithmetic — the two
    order statistics around index ``per/100 * (n-1)`` weighted in float64, ``sum(v * w) / sum(w)`` — but from a
    partition instead of a full sort.  ``np.percentile`` interpolates as ``a + (b - a) * t`` (in float32 for a float32
    image) and differs from it in the last bits, which moves threshold compares."""
    flat = np.asarray(a).ravel()
    if flat.size == 0:
        return np.nan
    if not (0 <= per <= 100):
        raise ValueError("percentile must be in the range [0, 100]")
    idx = per / 100. * (flat.size - 1)
    i = int(idx)
    if i == idx:
        return np.add.reduce(np.partition(flat, i)[i:i + 1] * np.array(1), axis=0) / 1.0
    w = np.array([(i + 1 - idx), (idx - i)], float)
    return np.add.reduce(np.partition(flat, [i, i + 1])[i:i + 2] * w, axis=0) / w.sum()


def remove_edge_points(im, T_seeds, distance=2):
    """spot_tools/fitting.py:156-165."""
    im_size = np.array(np.shape(im))
    _seeds = np.array(T_seeds)[:len(im_size), :].transpose()
    if len(_seeds) == 0:
        return np.zeros(0, dtype=bool)
    return ((_seeds >= distance) & (_seeds <= im_size - distance)).all(1)


def fit_fov_image(im, channel, seeds=None,
                  seed_mask=None,
                  max_num_seeds=500,
                  th_seed=300, th_seed_per=95, use_percentile=False,
                  use_dynamic_th=True,
                  dynamic_niters=10, min_dynamic_seeds=1,
                  remove_hot_pixel=True, seeding_kwargs={},
                  fit_radius=5,
                  normalize_background=False, normalize_local=False,
                  background_args={},
                  fitting_args={},
                  remove_boundary_points=True, verbose=True):
    """spot_tools/fitting.py:169-262 — seed + fit a whole field of view; (M,11) float32 rows
    [height,z,x,y,background,sigma_z,sigma_x,sigma_y,sin_t,sin_p,eps]."""
    th_seed = float(th_seed)
    if verbose:
        print(f"-- start fitting spots in channel:{channel}, ", end='')
        _fit_time = time.time()
    _resident = isinstance(im, L.DeviceStack)   # e.g. from correct_fov_image(..., return_device=True)
    if not _resident and not isinstance(im, np.ndarray):
        raise TypeError(f"image given should be a numpy.ndarray, but {type(im)} is given.")
    _stack = im if _resident else L.DeviceStack.upload(im)
    _shape = tuple(_stack.shape)
    try:
        if seeds is None:
            _seeds = _get_seeds_dev(_stack, max_num_seeds=max_num_seeds, th_seed=th_seed,
                                    th_seed_per=th_seed_per, use_percentile=use_percentile,
                                    use_dynamic_th=use_dynamic_th, dynamic_niters=dynamic_niters,
                                    min_dynamic_seeds=min_dynamic_seeds, remove_hot_pixel=remove_hot_pixel,
                                    host_im=None if _resident else im, **seeding_kwargs)
            if verbose:
                print(f"{len(_seeds)} seeded with th={th_seed}, ", end='')
        else:
            _seeds = np.array(seeds)[:, :len(_shape)]
            if verbose:
                print(f"{len(_seeds)} given, ", end='')
        if len(_seeds) == 0:
            return np.array([])
        if seed_mask is not None:                                            # :210-218
            _idx = np.round(_seeds[:, :len(_shape)]).astype(np.int32)
            _sel = seed_mask[tuple(_idx.T)] > 0
            _seeds = _seeds[_sel] if _sel.any() else np.array([])
            if verbose:
                print(f"{len(_seeds)} selected by mask, ", end='')
        _fitter = Fitting_v4.iter_fit_seed_points(_stack, np.asarray(_seeds).T, radius_fit=fit_radius,
                                                  **fitting_args)
        _fitter.firstfit()
        _fitter.repeatfit()
        _spots = np.array(_fitter.ps)
        _spots = _spots[np.sum(np.isnan(_spots), axis=1) == 0]               # :232
        if remove_boundary_points:                                           # :234-237
            _kept = (_spots[:, 1:4] > np.zeros(3)).all(1) * (_spots[:, 1:4] < np.array(_shape)).all(1)
            _spots = _spots[np.where(_kept)[0]]
        # intensity normalisation on the copy that is still resident (background.hip)
        if normalize_background and not normalize_local:                     # :240-245
            from ..io_tools.load import find_image_background
            _back = find_image_background(_stack, **background_args)
            if verbose:
                print(f"normalize total background:{_back:.2f}, ", end='')
            _spots[:, 0] = _spots[:, 0] / _back
        elif normalize_local:                                                # :246-258
            from ..io_tools.load import find_local_backgrounds
            _backs = find_local_backgrounds(_stack, _spots[:, 1:4], fit_radius * 2, **background_args)
            if verbose:
                print("normalize local background for each spot, ", end='')
            _spots[:, 0] = _spots[:, 0] / np.array(_backs)
    finally:
        if not _resident:
            _stack.free()
    if verbose:
        print(f"{len(_spots)} fitted in {time.time()-_fit_time:.3f}s.")
    return _spots


def fit_fov_images(ims, channels=None, n_workers=2, **kwargs):
    """``[fit_fov_image(im, ch, **kwargs) for im, ch in zip(ims, channels)]`` with ``n_workers`` images in flight.

    libia3 gives every host thread its own HIP streams, so independent images (the reference hands them to a process
    pool, classes/field_of_view.py:1129-1138) overlap on the device: while the fit kernel of one image drains its last
    few long-running fits, the filters and fits of the next images use the rest of the chip.  Results are identical to
    the sequential calls.  ``ims``: ndarrays or resident ``DeviceStack``s."""
    from concurrent.futures import ThreadPoolExecutor
    ims = list(ims)
    if channels is None:
        channels = [None] * len(ims)
    L.lib()   # load once, before the threads start
    if any(isinstance(_im, L.DeviceStack) for _im in ims):
        L.check(L.lib().ia3_sync())   # resident inputs may still be in production on this thread's stream
    kwargs.setdefault("verbose", False)
    # plain seed + fit on same-sized images: one ia3_fit_fovs call (library-owned threads and streams, no Python threads)
    _plain = {"th_seed", "max_num_seeds", "use_dynamic_th", "dynamic_niters", "min_dynamic_seeds", "remove_hot_pixel",
              "fit_radius", "verbose"}
    if (n_workers > 1 and len(ims) > 1 and set(kwargs) <= _plain
            and len({(tuple(_im.shape), np.dtype(_im.dtype).str) for _im in ims if hasattr(_im, "shape")}) == 1
            and all(isinstance(_im, (np.ndarray, L.DeviceStack)) and len(_im.shape) == 3 for _im in ims)):
        _sp, _keep = L.make_seed_params(float(kwargs.get("th_seed", 300)), max_num_seeds=kwargs.get("max_num_seeds", 500),
                                        use_dynamic_th=kwargs.get("use_dynamic_th", True),
                                        dynamic_niters=kwargs.get("dynamic_niters", 10),
                                        min_dynamic_seeds=kwargs.get("min_dynamic_seeds", 1),
                                        remove_hot_pixel=kwargs.get("remove_hot_pixel", True))
        _tables, _info = L.fit_fovs(ims, _sp, L.make_fit_params(radius_fit=kwargs.get("fit_radius", 5)),
                                    in_flight=n_workers)
        return [_t if _i["n_seeds"] else np.array([]) for _t, _i in zip(_tables, _info)]   # no seeds: fitting.py:206-207
    if n_workers <= 1 or len(ims) <= 1:
        return [fit_fov_image(_im, _ch, **kwargs) for _im, _ch in zip(ims, channels)]
    with ThreadPoolExecutor(max_workers=int(n_workers)) as pool:
        return list(pool.map(lambda a: fit_fov_image(a[0], a[1], **kwargs), zip(ims, channels)))


def _get_seeds_dev(stack, host_im=None, max_num_seeds=None, th_seed=150, th_seed_per=95, use_percentile=False,
                   sel_center=None, seed_radius=30, gfilt_size=0.75, background_gfilt_size=7.5, filt_size=3,
                   min_edge_distance=2, use_dynamic_th=True, dynamic_niters=10, min_dynamic_seeds=1,
                   remove_hot_pixel=True, hot_pixel_th=3, return_h=False, verbose=False):
    """get_seeds on a stack that is already resident in HBM (no second upload)."""
    if sel_center is not None or use_percentile:
        if host_im is None:
            host_im = stack.download()
        return get_seeds(host_im, max_num_seeds=max_num_seeds, th_seed=th_seed, th_seed_per=th_seed_per,
                         use_percentile=use_percentile, sel_center=sel_center, seed_radius=seed_radius,
                         gfilt_size=gfilt_size, background_gfilt_size=background_gfilt_size,
                         filt_size=filt_size, min_edge_distance=min_edge_distance,
                         use_dynamic_th=use_dynamic_th, dynamic_niters=dynamic_niters,
                         min_dynamic_seeds=min_dynamic_seeds, remove_hot_pixel=remove_hot_pixel,
                         hot_pixel_th=hot_pixel_th, return_h=return_h, verbose=verbose)
    if th_seed_per >= 100 or th_seed_per <= 50:
        print(f"th_seed_per should be a percentile > 50, invalid value given ({th_seed_per}), so not use percentile here.")
    _p, _keep = L.make_seed_params(th_seed, gfilt_size=gfilt_size, background_gfilt_size=background_gfilt_size,
                                   filt_size=filt_size, min_edge_distance=min_edge_distance,
                                   use_dynamic_th=use_dynamic_th, dynamic_niters=dynamic_niters,
                                   min_dynamic_seeds=min_dynamic_seeds, remove_hot_pixel=remove_hot_pixel,
                                   hot_pixel_th=hot_pixel_th, max_num_seeds=max_num_seeds)
    _cap = 16384
    while True:
        _out = np.empty((_cap, 4), dtype=np.float64)
        _n, _th_used = C.c_int(0), C.c_double(0)
        _rc = L.lib().ia3_dog_seed_dev(stack._h, C.byref(_p), L.dptr(_out), _cap, C.byref(_n), C.byref(_th_used))
        if _rc == L.IA3_ECAPACITY and _n.value > _cap:
            _cap = _n.value
            continue
        L.check(_rc)
        break
    _final = _out[:_n.value]
    return _final.copy() if return_h else _final[:, :3].copy()


def get_centers(im, seeds=None, th_seed=150,
                th_seed_per=98, use_percentile=False,
                sel_center=None, seed_radius=40,
                max_num_seeds=None, use_dynamic_th=True,
                min_num_seeds=1,
                remove_hot_pixel=True, hot_pixel_th=3,
                seed_kwargs={},
                fit_radius=5,
                remove_close_pts=True, close_threshold=0.1,
                verbose=False):
    """spot_tools/fitting.py:268-334 — fitted bead centres (K,3) float32: seeds (found here unless given), first fit and
    refit sweeps on the device, then (``remove_close_pts``) NaN rows, points with another point within
    ``close_threshold`` (squared distance) and points outside the image are dropped."""
    if seeds is None:
        seeding = dict(max_num_seeds=max_num_seeds, th_seed=th_seed, th_seed_per=th_seed_per,
                       use_percentile=use_percentile, sel_center=sel_center, seed_radius=seed_radius,
                       use_dynamic_th=use_dynamic_th, min_dynamic_seeds=min_num_seeds,
                       remove_hot_pixel=remove_hot_pixel, hot_pixel_th=hot_pixel_th, return_h=False, verbose=verbose)
        _twice = sorted(set(seeding) & set(seed_kwargs))
        if _twice:   # the reference passes both sets of keywords in one call
            raise TypeError(f"get_seeds() got multiple values for keyword argument '{_twice[0]}'")
        seeding.update(seed_kwargs)
        seeds = get_seeds(im, **seeding)
    fitter = Fitting_v4.iter_fit_seed_points(im, seeds.T, radius_fit=fit_radius)
    fitter.firstfit()
    fitter.repeatfit()
    rows = fitter.ps
    if len(rows) == 0:
        if verbose:
            print("-- no points fitted, return empty array.")
        return np.array([])
    centers = np.array(rows)[:, 1:4]
    if verbose:
        print(f"-- fitting {len(rows)} points.")
    if remove_close_pts:                                                     # :319-326
        crowded = np.zeros(len(centers), dtype=bool)
        for i0 in range(0, len(centers), 256):                               # row blocks: no n x n x 3 temporary
            sq = ((centers[i0:i0 + 256, None, :] - centers[None, :, :]) ** 2).sum(axis=-1)
            crowded[i0:i0 + 256] = (sq < close_threshold).sum(axis=1) > 1    # itself + at least one more
        outside = (centers < 0).any(axis=1) | (centers > np.array(im.shape)).any(axis=1)
        drop = np.isnan(centers).any(axis=1) | crowded | outside
        centers = centers[~drop]
        if verbose:
            print(f"-- {int(drop.sum())} points removed, given miminum distance {close_threshold}.")
    return centers


def select_sparse_centers(centers, distance_th=9,
                          distance_norm=np.inf,
                          verbose=False):
    """spot_tools/fitting.py:338-363 — greedy thinning in input order: a centre is kept when no centre kept before it
    lies within ``distance_th`` (norm ``distance_norm``).  Host logic on a few hundred rows."""
    kept = []
    for c in centers:
        if kept and (np.linalg.norm(np.array(kept) - c, axis=1, ord=distance_norm) <= distance_th).any():
            continue
        kept.append(c)
    if verbose:
        print(f"-- {len(kept)} among {len(centers)} centers are selected by th={distance_th}")
    return np.array(kept)
