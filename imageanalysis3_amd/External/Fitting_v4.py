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
        p = L.make_fit_params(self.radius_fit, self.min_delta_center, self.max_delta_center, self.n_max_iter,
                              self.max_dist_th, self.min_w, self.max_w, self.init_w)
        c = np.ascontiguousarray(self.centers, dtype=np.float64)
        h = C.c_void_p()
        L.check(L.lib().ia3_fit_create(stack._h, L.dptr(c), len(c), C.byref(p), C.byref(h)))
        self._fitter = h

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
