"""Drop-in for the hot-path part of the reference's ``External/Fitting_v4.py``:
``iter_fit_seed_points`` (:559-683) backed by the wave-per-ball LM kernels of libia3.so.

The class keeps the reference's constructor, ``firstfit()`` / ``repeatfit()`` and the attributes
callers read (``ps``, ``centers_fit``, ``success``, ``n_iter``, ``centers``).  The float64 residual
stack ``im_subtr`` / ``im_add`` and the per-seed ``ims_rec`` arrays of the reference are internal
state of its Gauss-Seidel loop; the device path never materialises them (fit.hip header).
"""
import ctypes as C
import numpy as np

from .. import _lib as L


def in_dim(x, y, z, xmax, ymax, zmax):
    """External/Fitting_v4.py:399-401."""
    keep = ((x >= 0) & (x < xmax) & (y >= 0) & (y < ymax) & (z >= 0) & (z < zmax)) > 0
    return x[keep], y[keep], z[keep]


def gaussfit_batch(ims, Xs, centers, delta_center=3., min_w=0.5, max_w=4., init_w=1.5):
    """Run many independent ``GaussianFit(im, X, center=...).fit()`` in one launch (one wave per fit).
    ims: list of 1-D voxel-value arrays; Xs: list of (3,n) coordinate arrays; centers: (N,3).
    Returns (ps (N,11) float32, xs (N,10) float64, success (N,) bool, nfev (N,) int)."""
    n_fits = len(ims)
    off = np.zeros(n_fits + 1, dtype=np.int32)
    kinds = np.zeros(n_fits, dtype=np.int32)
    for i, im in enumerate(ims):
        im = np.asarray(im)
        off[i + 1] = off[i] + im.size
        kinds[i] = 1 if im.dtype.kind in "ui" else (0 if im.dtype == np.float32 else 2)
    vals = np.concatenate([np.asarray(im, dtype=np.float64).ravel() for im in ims]) if n_fits else np.zeros(0)
    coords = (np.concatenate([np.asarray(X).T.reshape(-1, 3) for X in Xs]).astype(np.int32)
              if n_fits else np.zeros((0, 3), np.int32))
    vals, coords = np.ascontiguousarray(vals), np.ascontiguousarray(coords)
    cen = np.ascontiguousarray(np.asarray(centers, dtype=np.float64).reshape(n_fits, 3))
    cfg = np.ascontiguousarray(np.tile(np.array([delta_center, min_w, max_w, init_w], dtype=np.float64), (n_fits, 1)))
    ps = np.full((n_fits, 11), np.nan, dtype=np.float32)
    xs = np.full((n_fits, 10), np.nan, dtype=np.float64)
    info = np.zeros((n_fits, 2), dtype=np.int32)
    L.check(L.lib().ia3_gaussfit_voxels(L.dptr(vals), L.ptr(coords), L.ptr(off), n_fits, L.dptr(cen), L.dptr(cfg),
                                        L.ptr(kinds), L.ptr(ps), L.dptr(xs), L.ptr(info)))
    return ps, xs, info[:, 0].astype(bool), info[:, 1]


class GaussianFit():
    """External/Fitting_v4.py:165-396 — one constrained 10-parameter 3-D Gaussian fit on an explicit voxel
    list.  ``fit()`` runs the wave-per-fit LM kernel (ia3_gaussfit_voxels); the small helpers below
    (``to_natural_paramaters``, ``get_im``) evaluate the closed-form model on the host for the <= 512
    voxels of one fit, as the reference does."""

    def __init__(self, im, X, center=None, n_aprox=10, min_w=0.5, max_w=4., delta_center=3.,
                 init_w=1.5):
        self._min_w, self._max_w, self._init_w = min_w, max_w, init_w
        self.min_w = min_w * min_w
        self.max_w = max_w * max_w
        self.delta_center = delta_center
        self._im_in = np.asarray(im)
        self._X_in = np.asarray(X)
        self.im = np.array(im, dtype=np.float32)
        self.x, self.y, self.z = np.array(X, dtype=np.float32)
        argsort_im = np.argsort(im)
        if center is None:                                                        # :176-177
            center = np.median(self._X_in[:, argsort_im][:, -n_aprox:], -1)
        self.center_est = center
        if n_aprox != 10:
            raise NotImplementedError("n_aprox != 10")
        sorted_im = self._im_in[argsort_im]
        eps = np.exp(-10.)
        bk_guess = np.log(np.max([np.mean(sorted_im[:n_aprox]), eps]))
        h_guess = np.log(np.max([np.mean(sorted_im[-n_aprox:]), eps]))
        wsq = init_w ** 2
        wg = np.log((self.max_w - wsq) / (wsq - self.min_w))
        self.p_ = np.array([bk_guess, h_guess, 0, 0, 0, wg, wg, wg, 0, 0], dtype=np.float32)
        self.to_natural_paramaters()
        self.success = False

    # -- closed-form model on the host (Fitting_v4.py:189-290) ---------------------------------
    def _sig(self, v, lo, hi):
        v = np.float64(v)
        lim = np.log(np.finfo(np.float64).max)
        if v >= lim:
            return lo
        if v <= -lim:
            return hi
        return (hi - lo) / (1. + np.exp(v)) + lo

    def _geom(self, parms):
        bk, h, xp, yp, zp, w1, w2, w3, pp, tp = [np.float64(v) for v in parms]
        t, p = self._sig(tp, -1., 1.), self._sig(pp, -1., 1.)
        ws = [self._sig(w, self.min_w, self.max_w) for w in (w1, w2, w3)]
        d = self.delta_center
        c = [self._sig(v, -d, d) + np.float64(c0) for v, c0 in zip((xp, yp, zp), self.center_est)]
        return bk, h, t, p, ws, c

    def calc_f(self, parms):
        self.p_ = parms
        bk, h, t, p, (ws1, ws2, ws3), (xc, yc, zc) = self._geom(parms)
        xt, yt, zt = self.x - xc, self.y - yc, self.z - zc
        p2, t2 = p * p, t * t
        tc2, pc2 = 1 - t2, 1 - p2
        tc, pc = np.sqrt(tc2), np.sqrt(pc2)
        s1, s2, s3 = 1. / ws1, 1. / ws2, 1. / ws3
        x2c = pc2 * tc2 * s1 + t2 * s2 + p2 * tc2 * s3
        y2c = pc2 * t2 * s1 + tc2 * s2 + p2 * t2 * s3
        z2c = p2 * s1 + pc2 * s3
        xyc = 2 * tc * t * (pc2 * s1 - s2 + p2 * s3)
        xzc = 2 * p * pc * tc * (s3 - s1)
        yzc = 2 * p * pc * t * (s3 - s1)
        xsigmax = x2c * xt * xt + y2c * yt * yt + z2c * zt * zt + xyc * xt * yt + xzc * xt * zt + yzc * yt * zt
        self.f0 = np.exp(h - 0.5 * xsigmax)
        self.f = np.exp(np.clip(bk, -709.78, 709.78)) + self.f0
        return self.f

    def calc_eps(self, parms):
        return self.calc_f(parms) - self.im

    def to_natural_paramaters(self, parms=None):
        if parms is None:
            parms = self.p_
        bk, h, t, p, ws, c = self._geom(parms)
        eps = np.mean(np.abs(self.calc_eps(parms)))
        self.p = np.array([np.exp(h), c[0], c[1], c[2], np.exp(bk), np.sqrt(ws[0]), np.sqrt(ws[1]), np.sqrt(ws[2]),
                           t, p, eps], dtype=np.float32)
        return self.p

    def fit(self, eps_frac=10E-3, eps_dist=10E-3, eps_angle=10E-3):
        """Levenberg-Marquardt on the device; results in ``self.p`` = [height, c0, c1, c2, background,
        width_0, width_1, width_2, sin_theta, sin_phi, error] (Fitting_v4.py:377-393)."""
        if len(self.p_) > len(self.im):
            self.success = False
        else:
            ps, xs, ok, nfev = gaussfit_batch([self._im_in], [self._X_in], [self.center_est],
                                              delta_center=self.delta_center, min_w=self._min_w,
                                              max_w=self._max_w, init_w=self._init_w)
            self.p_ = xs[0]
            self.p = ps[0]
            self.center = self.p[1:4]
            self.nfev = int(nfev[0])
            self.success = True

    def get_im(self):
        self.calc_f(self.p_)
        return self.f0


class iter_fit_seed_points():
    def __init__(self, im, centers, radius_fit=5, min_delta_center=1., max_delta_center=2.5,
                 n_max_iter=10, max_dist_th=0.1,
                 min_w=0.5, max_w=4, init_w=1.5):
        """``im``: (z,x,y) ndarray (uint16/float32) or a DeviceStack; ``centers``: (3,N) like the
        reference (it stores ``centers.T``)."""
        self.im = im
        self.radius_fit = radius_fit
        self.n_max_iter = n_max_iter
        self.max_dist_th = max_dist_th
        self.min_delta_center = min_delta_center
        self.max_delta_center = max_delta_center
        centers = np.asarray(centers, dtype=np.float64)
        self.centers = centers.T if centers.size else np.zeros((0, 3))
        if self.centers.ndim != 2 or (len(self.centers) and self.centers.shape[1] != 3):
            raise IndexError("centers should be a (3, N) array")
        self.z, self.x, self.y = (self.centers[:, 0], self.centers[:, 1], self.centers[:, 2])
        self.zb, self.xb, self.yb = np.reshape(np.indices([self.radius_fit * 2] * 3) - self.radius_fit, [3, -1])
        keep = self.zb * self.zb + self.xb * self.xb + self.yb * self.yb <= self.radius_fit ** 2
        self.zb, self.xb, self.yb = self.zb[keep], self.xb[keep], self.yb[keep]
        self.zxyb = np.array([self.zb, self.xb, self.yb]).T
        self.sz, self.sx, self.sy = im.shape
        self.min_w = min_w
        self.max_w = max_w
        self.init_w = init_w
        self._own_stack = None
        self._fitter = None
        self.ps = []
        self.success = []
        self.centers_fit = []
        self.n_iter = 0

    # -- device plumbing ---------------------------------------------------------------------
    def _ensure(self):
        if self._fitter is not None:
            return
        if isinstance(self.im, L.DeviceStack):
            stack = self.im
        else:
            self._own_stack = L.DeviceStack.upload(self.im)
            stack = self._own_stack
        self._stack = stack
        p = self._fit_params()
        c = np.ascontiguousarray(self.centers, dtype=np.float64)
        h = C.c_void_p()
        L.check(L.lib().ia3_fit_create(stack._h, L.dptr(c), len(c), C.byref(p), C.byref(h)))
        self._fitter = h

    def _fit_params(self):
        return L.make_fit_params(self.radius_fit, self.min_delta_center, self.max_delta_center, self.n_max_iter,
                                 self.max_dist_th, self.min_w, self.max_w, self.init_w)

    def _pull(self):
        n = len(self.centers)
        ps = np.empty((n, 11), dtype=np.float32)
        ok = np.empty(n, dtype=np.uint8)
        nv = np.empty(n, dtype=np.int32)
        L.check(L.lib().ia3_fit_results(self._fitter, L.ptr(ps), L.ptr(ok), L.ptr(nv)))
        self.ps = [ps[i] for i in range(n)]
        self.success = [bool(v) for v in ok]
        self.centers_fit = [ps[i, 1:4] for i in range(n)]
        self.nvox = nv

    def _release(self):
        if self._fitter is not None:
            L.lib().ia3_fit_destroy(self._fitter)
            self._fitter = None
        if self._own_stack is not None:
            self._own_stack.free()
            self._own_stack = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # -- reference API -----------------------------------------------------------------------
    def firstfit(self):
        """External/Fitting_v4.py:590-639 — Voronoi-restricted first fit of every seed."""
        if len(self.centers) > 0:
            self._ensure()
            L.check(L.lib().ia3_fit_first(self._fitter))
            self._pull()

    def repeatfit(self):
        """External/Fitting_v4.py:641-683 — ordered Gauss-Seidel refit sweeps until converged."""
        self.n_iter = 0
        self.converged = np.zeros(len(self.centers), dtype=bool)
        if len(self.centers) > 0:
            if self._fitter is None:
                raise AttributeError("repeatfit() called before firstfit()")
            n_iter = C.c_int(0)
            L.check(L.lib().ia3_fit_repeat(self._fitter, C.byref(n_iter)))
            self.n_iter = int(n_iter.value)
            self._pull()
            self.converged[:] = True
            self._release()

    def stats(self):
        """(number of LM fits run, total function evaluations) so far — for the flop accounting."""
        a, b = C.c_int64(0), C.c_int64(0)
        L.check(L.lib().ia3_fit_stats(self._fitter, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)
