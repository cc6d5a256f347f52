"""ctypes binding of libia3.so (C ABI declared in include/ia3.h).

The library is the product: there is no CPU fallback.  Importing this module without a built
``libia3.so`` raises ImportError; calling into it without a HIP device raises RuntimeError.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IA3_LIB_PATH") or os.path.join(_HERE, "libia3.so")   # IA3_LIB_PATH: developer builds (scripts/ab_*)

IA3_U16, IA3_F32 = 0, 1
IA3_OK, IA3_EINVAL, IA3_EHIP, IA3_ENOMEM, IA3_ECAPACITY, IA3_EUNSUPPORTED = 0, -1, -2, -3, -4, -5
MODE_REFLECT, MODE_NEAREST, MODE_CONSTANT = 0, 1, 2

EXPORTS = [
    "ia3_init", "ia3_last_error", "ia3_version", "ia3_device_name", "ia3_sync", "ia3_stream",
    "ia3_release_workspace", "ia3_workspace_stats", "ia3_prepare_depth", "ia3_profile_enable", "ia3_profile_collect", "ia3_set_tuning",
    "ia3_stack_upload", "ia3_stack_alloc", "ia3_stack_load_file", "ia3_stack_wrap", "ia3_stack_download", "ia3_stack_info",
    "ia3_stack_free",
    "ia3_gaussian_filter", "ia3_gaussian_filter_dev", "ia3_gaussian_highpass", "ia3_gaussian_highpass_dev",
    "ia3_remove_hot_pixels", "ia3_z_shift_correction", "ia3_illumination_correct", "ia3_bleedthrough_correct",
    "ia3_dog_seed", "ia3_dog_seed_dev", "ia3_dog_filters_dev", "ia3_seed_in_distance",
    "ia3_find_background", "ia3_find_background_dev", "ia3_local_background_dev",
    "ia3_stack_deinterleave", "ia3_buffer_upload", "ia3_buffer_free", "ia3_remove_hot_pixels_dev",
    "ia3_z_shift_correction_dev", "ia3_illumination_correct_dev", "ia3_bleedthrough_correct_dev",
    "ia3_illumination_rescale_dev", "ia3_bleedthrough_rescale_dev",
    "ia3_fit_create", "ia3_fit_first", "ia3_fit_repeat", "ia3_fit_run", "ia3_fit_results", "ia3_fit_results_ex", "ia3_fit_nfev", "ia3_fit_stats", "ia3_fit_counters",
    "ia3_fit_destroy", "ia3_fit_seeds", "ia3_fit_fov_dev", "ia3_fit_fov_stats", "ia3_fit_fov_wait_share", "ia3_fit_fovs",
    "ia3_gaussfit_voxels",
    "ia3_fftalign_2d", "ia3_fft3d_from2d", "ia3_fft3d_from2d_dev", "ia3_phase_xcorr3d", "ia3_phase_xcorr3d_dev",
    "ia3_stack_crop", "ia3_warp3d", "ia3_warp3d_dev",
    "ia3_align_image_dev", "ia3_process_movies", "ia3_drift_ref_create", "ia3_drift_ref_free", "ia3_align_image_ref",
]


class SeedParams(C.Structure):
    _fields_ = [("th_seed", C.c_double), ("gfilt_size", C.c_double), ("background_gfilt_size", C.c_double),
                ("filt_size", C.c_int), ("min_edge_distance", C.c_int), ("use_dynamic_th", C.c_int),
                ("dynamic_niters", C.c_int), ("min_dynamic_seeds", C.c_int), ("remove_hot_pixel", C.c_int),
                ("hot_pixel_th", C.c_int), ("max_num_seeds", C.c_int), ("th_compare_f32", C.c_int),
                ("w_front", C.POINTER(C.c_double)), ("r_front", C.c_int),
                ("w_back", C.POINTER(C.c_double)), ("r_back", C.c_int)]


class LegacySeedParams(C.Structure):
    _fields_ = [("num_seeds", C.c_int), ("seed_radius", C.c_double), ("gfilt_size", C.c_double),
                ("background_gfilt_size", C.c_double), ("filt_size", C.c_int), ("th_seed", C.c_double),
                ("dynamic", C.c_int), ("dynamic_iters", C.c_int), ("min_dynamic_seeds", C.c_int),
                ("hot_pix_th", C.c_int)]


class FitParams(C.Structure):
    _fields_ = [("radius_fit", C.c_int), ("min_delta_center", C.c_double), ("max_delta_center", C.c_double),
                ("n_max_iter", C.c_int), ("max_dist_th", C.c_double), ("min_w", C.c_double),
                ("max_w", C.c_double), ("init_w", C.c_double),
                ("model_variant", C.c_int), ("init_w_zxy", C.c_double * 3)]


class FovJob(C.Structure):
    _fields_ = [("host", C.c_void_p), ("dev", C.c_void_p), ("rows", C.c_void_p), ("capacity", C.c_int),
                ("n_rows", C.c_int), ("n_seeds", C.c_int), ("n_iter", C.c_int), ("rc", C.c_int),
                ("fits", C.c_longlong), ("nfev", C.c_longlong), ("voxel_evals", C.c_longlong)]


MOVIE_MAXCH = 8


class MovieParams(C.Structure):
    """ia3_movie_params (include/ia3.h)."""
    _fields_ = [("frames", C.c_int), ("X", C.c_int), ("Y", C.c_int), ("Z", C.c_int),
                ("n_load", C.c_int), ("load_start", C.c_int * MOVIE_MAXCH), ("load_step", C.c_int),
                ("n_sel", C.c_int), ("sel", C.c_int * MOVIE_MAXCH),
                ("hot_pixel_corr", C.c_int), ("hot_pixel_th", C.c_double), ("z_shift_corr", C.c_int),
                ("n_bleed", C.c_int), ("bleed_idx", C.c_int * MOVIE_MAXCH),
                ("bleed_profile", C.c_void_p), ("bleed_dtype", C.c_int),
                ("illum_profile", C.c_void_p * MOVIE_MAXCH), ("illum_dtype", C.c_int * MOVIE_MAXCH),
                ("drift_idx", C.c_int), ("ref_bead", C.c_void_p), ("drift_ref", C.c_void_p),
                ("n_crops", C.c_int), ("crops", C.c_int * 48),
                ("precision_fold", C.c_int), ("normalization", C.c_int), ("min_good_drifts", C.c_int),
                ("drift_diff_th", C.c_double),
                ("warp", C.c_int), ("warp_always", C.c_int * MOVIE_MAXCH),
                ("chrom_field", C.c_void_p * MOVIE_MAXCH), ("chrom_dtype", C.c_int * MOVIE_MAXCH),
                ("highpass_sigma", C.c_double), ("highpass_truncate", C.c_double),
                ("fit_spots", C.c_int),
                ("seed", SeedParams * MOVIE_MAXCH), ("fit", FitParams),
                ("normalize", C.c_int), ("bg_crop_size", C.c_int), ("bg_edges", C.POINTER(C.c_double)),
                ("bg_n_edges", C.c_int), ("bg_max_iter", C.c_int),
                ("correct_threads", C.c_int), ("fit_group_images", C.c_int), ("upload_ahead", C.c_int)]


class MovieJob(C.Structure):
    """ia3_movie_job (include/ia3.h)."""
    _fields_ = [("host_raw", C.c_void_p), ("path", C.c_char_p), ("offset_bytes", C.c_longlong), ("big_endian", C.c_int),
                ("drift_in", C.c_double * 3), ("measure_drift", C.c_int),
                ("images_out", C.c_void_p * MOVIE_MAXCH),
                ("rows", C.c_void_p * MOVIE_MAXCH), ("capacity", C.c_int * MOVIE_MAXCH),
                ("drift", C.c_double * 3), ("drift_flag", C.c_int),
                ("n_rows", C.c_int * MOVIE_MAXCH), ("n_seeds", C.c_int * MOVIE_MAXCH), ("n_iter", C.c_int * MOVIE_MAXCH),
                ("rc", C.c_int),
                ("t_upload_ms", C.c_double), ("t_correct_ms", C.c_double), ("t_fit_ms", C.c_double),
                ("stamps", C.c_double * 6)]


_lib = None


def lib():
    """Load libia3.so (once).  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise ImportError(
                "imageanalysis3_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C imageanalysis3_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        # streams of concurrent host threads onto separate hardware queues (see runtime.cpp do_init); must be in the
        # environment before the HIP runtime starts, hence also here, before anything is loaded
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
        L = C.CDLL(LIB_PATH)
        L.ia3_last_error.restype = C.c_char_p
        L.ia3_version.restype = C.c_char_p
        L.ia3_stream.restype = C.c_void_p
        L.ia3_stack_free.restype = None
        L.ia3_fit_destroy.restype = None
        L.ia3_drift_ref_free.restype = None
        _lib = L
    return _lib


class IA3Error(RuntimeError):
    pass


def check(rc, allow=()):
    if rc == IA3_OK or rc in allow:
        return rc
    msg = lib().ia3_last_error().decode("utf-8", "replace")
    if rc == IA3_EINVAL:
        raise ValueError(msg)
    if rc == IA3_ENOMEM:
        raise MemoryError(msg)
    if rc == IA3_EUNSUPPORTED:
        raise NotImplementedError(msg)
    raise IA3Error("libia3 error %d: %s" % (rc, msg))


def dtype_code(arr):
    if arr.dtype == np.uint16:
        return IA3_U16
    if arr.dtype == np.float32:
        return IA3_F32
    raise TypeError("imageanalysis3_amd kernels take uint16 or float32 stacks, got %s" % arr.dtype)


def as_stack_array(im):
    """C-contiguous uint16/float32 3-D view/copy of ``im`` (the caller's array is never modified)."""
    if not isinstance(im, np.ndarray):
        raise TypeError("image given should be a numpy.ndarray, but %s is given." % type(im))
    if im.ndim != 3:
        raise IndexError("a 3-D (z,x,y) stack is required, got ndim=%d" % im.ndim)
    if im.dtype == np.uint8:
        im = im.astype(np.uint16)
    dtype_code(im)
    return np.ascontiguousarray(im)


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class DeviceStack(object):
    """A (Z,X,Y) stack resident in HBM (owner of an ``ia3_stack`` handle)."""

    def __init__(self, handle, shape, dtype, keepalive=None):
        self._h = handle
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self._keep = keepalive

    @classmethod
    def upload(cls, im):
        a = as_stack_array(im)
        h = C.c_void_p()
        check(lib().ia3_stack_upload(ptr(a), dtype_code(a), a.shape[0], a.shape[1], a.shape[2], C.byref(h)))
        return cls(h, a.shape, a.dtype)

    @classmethod
    def empty(cls, shape, dtype):
        dt = np.dtype(dtype)
        code = dtype_code(np.empty(0, dt))
        h = C.c_void_p()
        check(lib().ia3_stack_alloc(code, int(shape[0]), int(shape[1]), int(shape[2]), C.byref(h)))
        return cls(h, shape, dt)

    @classmethod
    def wrap_torch(cls, t):
        """Borrow the memory of a contiguous CUDA/HIP torch tensor (uint16 as int16/uint16, or float32)."""
        import torch
        if not t.is_cuda or not t.is_contiguous() or t.dim() != 3:
            raise ValueError("need a contiguous 3-D device tensor")
        if t.dtype == torch.float32:
            dt = np.float32
        elif t.dtype in (torch.int16, getattr(torch, "uint16", torch.int16)):
            dt = np.uint16
        else:
            raise TypeError("unsupported tensor dtype %s" % t.dtype)
        h = C.c_void_p()
        code = IA3_F32 if dt == np.float32 else IA3_U16
        check(lib().ia3_stack_wrap(C.c_void_p(t.data_ptr()), code, t.shape[0], t.shape[1], t.shape[2], C.byref(h)))
        return cls(h, tuple(t.shape), dt, keepalive=t)

    @classmethod
    def from_file(cls, path, frames, X, Y, offset_bytes=0, big_endian=False):
        """Resident uint16 (frames, X, Y) stack read straight from a raw movie file (pipelined read + upload)."""
        import os
        h = C.c_void_p()
        check(lib().ia3_stack_load_file(os.fsencode(path), C.c_longlong(int(offset_bytes)), int(frames), int(X), int(Y),
                                        1 if big_endian else 0, C.byref(h)))
        return cls(h, (int(frames), int(X), int(Y)), np.dtype(np.uint16))

    def crop(self, lims):
        """New resident stack = self[z0:z1, x0:x1, y0:y1]; ``lims`` is a (3,2) [start, stop) array."""
        l = np.array(lims, dtype=int)
        l[:, 0] = np.maximum(l[:, 0], 0)
        l[:, 1] = np.minimum(l[:, 1], np.array(self.shape))
        h = C.c_void_p()
        check(lib().ia3_stack_crop(self._h, int(l[0, 0]), int(l[0, 1]), int(l[1, 0]), int(l[1, 1]),
                                   int(l[2, 0]), int(l[2, 1]), C.byref(h)))
        return DeviceStack(h, tuple(int(b - a) for a, b in l), self.dtype)

    def download(self):
        out = np.empty(self.shape, dtype=self.dtype)
        check(lib().ia3_stack_download(self._h, ptr(out)))
        return out

    def free(self):
        if self._h is not None:
            lib().ia3_stack_free(self._h)
            self._h = None
            self._keep = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.free()


def profile_enable(on=True):
    check(lib().ia3_profile_enable(1 if on else 0))


def profile_collect():
    """{kernel: (launches, total_ms)} measured with HIP events on the library stream since the last call."""
    buf = C.create_string_buffer(1 << 16)
    check(lib().ia3_profile_collect(buf, len(buf)))
    out = {}
    for line in buf.value.decode().splitlines():
        name, n, ms = line.rsplit(",", 2)
        out[name] = (int(n), float(ms))
    return out


def gaussian_taps(sigma, truncate=4.0):
    """scipy.ndimage._filters._gaussian_kernel1d(order=0) — the taps SciPy would use, from NumPy on
    this host, handed to the kernels verbatim."""
    sigma = float(sigma)
    radius = int(float(truncate) * sigma + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return np.ascontiguousarray(phi / phi.sum(), dtype=np.float64), radius


def make_seed_params(th_seed, gfilt_size=0.75, background_gfilt_size=7.5, filt_size=3, min_edge_distance=2,
                     use_dynamic_th=True, dynamic_niters=10, min_dynamic_seeds=1, remove_hot_pixel=True,
                     hot_pixel_th=3, max_num_seeds=None):
    """Build an ia3_seed_params (+ the arrays it points to, which the caller must keep alive)."""
    keep = []
    p = SeedParams()
    p.th_seed = float(th_seed)
    # NumPy >= 2 promotion: a Python float/int threshold is a weak scalar (compare in float32), a
    # NumPy float64 scalar is strong (compare in float64); float32/16 scalars compare in float32.
    strong64 = isinstance(th_seed, np.floating) and np.dtype(type(th_seed)).itemsize >= 8
    p.th_compare_f32 = 0 if strong64 else 1
    p.gfilt_size = float(gfilt_size) if gfilt_size else 0.0
    p.background_gfilt_size = float(background_gfilt_size) if background_gfilt_size else 0.0
    p.filt_size = int(filt_size)
    p.min_edge_distance = int(np.ceil(min_edge_distance)) if min_edge_distance > 0 else 0
    p.use_dynamic_th = 1 if use_dynamic_th else 0
    p.dynamic_niters = int(dynamic_niters)
    p.min_dynamic_seeds = int(min_dynamic_seeds)
    p.remove_hot_pixel = 1 if remove_hot_pixel else 0
    p.hot_pixel_th = int(hot_pixel_th)
    p.max_num_seeds = int(max_num_seeds) if (max_num_seeds is not None and max_num_seeds > 0) else 0
    if p.gfilt_size > 0:
        w, r = gaussian_taps(p.gfilt_size)
        keep.append(w)
        p.w_front, p.r_front = dptr(w), r
    if p.background_gfilt_size > 0:
        w, r = gaussian_taps(p.background_gfilt_size)
        keep.append(w)
        p.w_back, p.r_back = dptr(w), r
    return p, keep


def make_fit_params(radius_fit=5, min_delta_center=1., max_delta_center=2.5, n_max_iter=10, max_dist_th=0.1,
                    min_w=0.5, max_w=4, init_w=1.5, model_variant=0, init_w_zxy=(1.35, 1.9, 1.9)):
    p = FitParams()
    p.model_variant = int(model_variant)
    for k in range(3):
        p.init_w_zxy[k] = float(init_w_zxy[k])
    p.radius_fit = int(radius_fit)
    p.min_delta_center = float(min_delta_center)
    p.max_delta_center = float(max_delta_center)
    p.n_max_iter = int(n_max_iter)
    p.max_dist_th = float(max_dist_th)
    p.min_w, p.max_w, p.init_w = float(min_w), float(max_w), float(init_w)
    return p


def fit_fovs(ims, seed_params, fit_params, in_flight=4, capacity=16384):
    """``ia3_fit_fovs`` on a list of same-shape, same-dtype images (ndarrays and/or resident ``DeviceStack``s):
    list of (M,11) float32 tables in input order + per-image dicts (n_seeds, n_iter, fits, nfev, voxel_evals).
    One C call; uploads, filters and fits of different images overlap on library-owned threads and streams."""
    ims = list(ims)
    if not ims:
        return [], []
    arrs = [im if isinstance(im, DeviceStack) else as_stack_array(im) for im in ims]
    shape, dt = tuple(arrs[0].shape), np.dtype(arrs[0].dtype)
    for a in arrs:
        if tuple(a.shape) != shape or np.dtype(a.dtype) != dt:
            raise ValueError("fit_fovs: every image of a batch must have the same shape and dtype")
    code = IA3_F32 if dt == np.float32 else IA3_U16
    n = len(arrs)
    while True:
        jobs = (FovJob * n)()
        rows = [np.empty((capacity, 11), dtype=np.float32) for _ in range(n)]
        for j, a, r in zip(jobs, arrs, rows):
            if isinstance(a, DeviceStack):
                j.dev = a._h
            else:
                j.host = a.ctypes.data
            j.rows, j.capacity = r.ctypes.data, capacity
        rc = lib().ia3_fit_fovs(jobs, n, code, shape[0], shape[1], shape[2], C.byref(seed_params), C.byref(fit_params),
                                int(in_flight))
        need = max(j.n_rows for j in jobs)
        if rc == IA3_ECAPACITY and need > capacity:   # a row table was too small (the seed stage reports the same code
            capacity = need                           # for "too many candidates", with n_rows = 0: that one is an error)
            continue
        check(rc)
        break
    tables = [r[:j.n_rows].copy() for j, r in zip(jobs, rows)]
    info = [dict(n_seeds=j.n_seeds, n_iter=j.n_iter, fits=j.fits, nfev=j.nfev, voxel_evals=j.voxel_evals) for j in jobs]
    return tables, info


def process_movies(params, movies, drifts_in=None, measure_drift=True, want_images=False, capacity=16384):
    """``ia3_process_movies``: ``movies`` = raw (frames, X, Y) uint16 arrays and/or .dax paths (``(path, offset_bytes,
    big_endian)`` tuples or plain strings), all of the layout ``params`` (a filled ``MovieParams``) describes.  Returns
    one dict per movie: ``tables`` (list of (M,11) float32 per selected channel), ``drift``, ``drift_flag``, ``images``
    (list of (Z,X,Y) uint16 or None), ``n_seeds``, ``n_iter``, ``ms`` (host wall time per stage).  ``drifts_in``: per movie
    a drift to use (or start from); ``measure_drift``: bool or per-movie list."""
    n = len(movies)
    if n == 0:
        return []
    n_sel = int(params.n_sel)
    Z, X, Y = int(params.Z), int(params.X), int(params.Y)
    keep = []
    while True:
        jobs = (MovieJob * n)()
        rows = [[np.empty((capacity, 11), dtype=np.float32) for _ in range(n_sel)] for _ in range(n)]
        images = [[np.empty((Z, X, Y), dtype=np.uint16) if want_images else None for _ in range(n_sel)] for _ in range(n)]
        for k, (j, m) in enumerate(zip(jobs, movies)):
            if isinstance(m, np.ndarray):
                if m.dtype != np.uint16 or m.ndim != 3 or tuple(m.shape) != (int(params.frames), X, Y):
                    raise TypeError("the raw movie should be a (%d, %d, %d) uint16 array" % (int(params.frames), X, Y))
                a = np.ascontiguousarray(m)
                keep.append(a)
                j.host_raw = a.ctypes.data
            else:
                path, off, big = (m, 0, False) if isinstance(m, (str, bytes)) else m
                j.path = os.fsencode(path)
                j.offset_bytes, j.big_endian = int(off), 1 if big else 0
            md = measure_drift[k] if isinstance(measure_drift, (list, tuple)) else measure_drift
            j.measure_drift = 1 if md else 0
            if drifts_in is not None and drifts_in[k] is not None:
                for a_ in range(3):
                    j.drift_in[a_] = float(drifts_in[k][a_])
            for s_ in range(n_sel):
                j.rows[s_], j.capacity[s_] = rows[k][s_].ctypes.data, capacity
                if want_images:
                    j.images_out[s_] = images[k][s_].ctypes.data
        rc = lib().ia3_process_movies(jobs, n, C.byref(params))
        need = max(max(j.n_rows[s_] for s_ in range(n_sel)) for j in jobs)
        if rc == IA3_ECAPACITY and need > capacity:
            capacity = need
            continue
        check(rc)
        break
    out = []
    for k, j in enumerate(jobs):
        out.append(dict(tables=[rows[k][s_][:j.n_rows[s_]].copy() for s_ in range(n_sel)],
                        drift=np.array([j.drift[0], j.drift[1], j.drift[2]]), drift_flag=int(j.drift_flag),
                        images=images[k] if want_images else None,
                        n_seeds=[int(j.n_seeds[s_]) for s_ in range(n_sel)], n_iter=[int(j.n_iter[s_]) for s_ in range(n_sel)],
                        ms=dict(upload=j.t_upload_ms, correct=j.t_correct_ms, fit=j.t_fit_ms),
                        stamps=[float(v) for v in j.stamps]))
    return out
