"""TEST INFRASTRUCTURE — CPU restatement (NumPy) of ImageAnalysis3's per-FOV spot-calling path.

This module is the *oracle*: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product package
(``imageanalysis3_amd``) never does — it fails loudly when the HIP library is missing.

Parity status: PINNED.  Every function below is checked in ``tests/test_oracle_golden.py``
against fixtures produced by running the reference's own Python in the development
container (``oracle/make_golden.py`` → ``tests/golden/*.npz``), and — where the reference tree
is present — directly against the reference (``tests/test_oracle_vs_reference.py``).
``phase_cross_correlation`` (scikit-image, not vendored by the reference, version unpinned, absent from the
system interpreter) is pinned against the real scikit-image 0.18.3 of the image's Anaconda interpreter
(``oracle/make_golden_h5.py phase`` → ``tests/golden/phase.npz``) for ``normalization=None``; the
``normalization="phase"`` flavour of scikit-image >= 0.19 is a restatement of the published algorithm validated by
known-answer tests only.

Third-party arithmetic the reference delegates to and this oracle keeps delegating to:
``scipy.optimize.leastsq`` (MINPACK lmder), ``scipy.spatial.cKDTree`` / ``Delaunay``,
``scipy.signal.fftconvolve``.  ``scipy.ndimage`` filters are restated in NumPy with the exact
summation order of ``NI_Correlate1D`` so the HIP kernels have a line-by-line model.

All ``file:line`` citations are relative to /root/reference/.
"""
import numpy as np

# ----------------------------------------------------------------------------------------------
# scipy.ndimage restatements
# ----------------------------------------------------------------------------------------------


def gaussian_kernel1d(sigma, truncate=4.0):
    """scipy.ndimage._filters._gaussian_kernel1d(order=0): radius=int(truncate*sigma+0.5)."""
    sigma = float(sigma)
    radius = int(truncate * sigma + 0.5)
    sigma2 = sigma * sigma
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / sigma2 * x ** 2)
    return phi / phi.sum(), radius


def _store(acc, dtype):
    """double -> output dtype exactly as NI's C casts do (floats round, ints truncate)."""
    dtype = np.dtype(dtype)
    if dtype.kind == "f":
        return acc.astype(dtype)
    return np.trunc(acc).astype(dtype)  # (npy_uint16)double : truncation toward zero


def correlate1d(a, weights, axis, mode="reflect"):
    """NI_Correlate1D, symmetric branch: out = in[0]*w0 + sum_{j=r..1} (in[-j]+in[+j])*w[j].

    float64 accumulation in exactly that order, result stored to ``a.dtype``.
    mode 'reflect' = half-sample symmetric (np.pad 'symmetric', multi-reflection safe),
    'nearest' = edge replicate.
    """
    r = (len(weights) - 1) // 2
    pad_mode = {"reflect": "symmetric", "nearest": "edge"}[mode]
    a = np.asarray(a)
    x = np.moveaxis(a, axis, 0).astype(np.float64)
    n = x.shape[0]
    xp = np.pad(x, [(r, r)] + [(0, 0)] * (x.ndim - 1), mode=pad_mode)
    acc = xp[r:r + n] * weights[r]
    for j in range(r, 0, -1):
        acc += (xp[r - j:r - j + n] + xp[r + j:r + j + n]) * weights[r - j]
    return np.moveaxis(_store(acc, a.dtype), 0, axis)


def gaussian_filter(a, sigma, mode="reflect", truncate=4.0):
    """scipy.ndimage.gaussian_filter: axes 0,1,2 in order, re-quantised to a.dtype per axis."""
    w, _ = gaussian_kernel1d(sigma, truncate)
    out = np.asarray(a)
    for ax in range(out.ndim):
        out = correlate1d(out, w, ax, mode)
    return out


def _rank3(a, fn):
    """3-wide max/min filter along every axis, mode 'reflect' (== edge clamp for size 3)."""
    out = np.asarray(a)
    for ax in range(out.ndim):
        p = np.pad(out, [(1, 1) if i == ax else (0, 0) for i in range(out.ndim)], mode="edge")
        sl = [slice(None)] * out.ndim
        parts = []
        for s in (slice(0, -2), slice(1, -1), slice(2, None)):
            sl[ax] = s
            parts.append(p[tuple(sl)])
        out = fn(fn(parts[0], parts[1]), parts[2])
    return out


def maximum_filter3(a):
    return _rank3(a, np.maximum)


def minimum_filter3(a):
    return _rank3(a, np.minimum)


# ----------------------------------------------------------------------------------------------
# (a6) correction_tools/filter.py:14-19
# ----------------------------------------------------------------------------------------------


def gaussian_high_pass_filter(image, sigma=5, truncate=2):
    """filter.py:14-19 — lowpass(mode nearest); out = image - lowpass; out[lowpass > image] = 0."""
    lowpass = gaussian_filter(image, sigma, mode="nearest", truncate=truncate)
    hp = image - lowpass  # wraps for unsigned ints, then zeroed
    hp[lowpass > image] = 0
    return hp


# ----------------------------------------------------------------------------------------------
# (a7) correction_tools/filter.py:22-42
# ----------------------------------------------------------------------------------------------


def remove_hot_pixels(im, dtype=np.uint16, hot_pix_th=0.50, hot_th=4):
    """filter.py:22-42 (note the duplicated np.roll(im,1,2) at :28 is reproduced)."""
    conv = (np.roll(im, 1, 1) + np.roll(im, -1, 1) + np.roll(im, 1, 2) + np.roll(im, 1, 2)) / 4
    hot = im > hot_th * conv
    hot2d = np.sum(hot, 0)
    cand = np.where(hot2d > hot_pix_th * im.shape[0])
    if len(cand[0]) == 0:
        return im
    nim = im.copy()
    for x, y in zip(cand[0], cand[1]):
        if 0 < x < im.shape[1] - 1 and 0 < y < im.shape[2] - 1:
            nim[:, x, y] = (nim[:, x + 1, y] + nim[:, x - 1, y] + nim[:, x, y + 1] + nim[:, x, y - 1]) / 4
    return nim.astype(dtype)


# ----------------------------------------------------------------------------------------------
# (a1) spot_tools/fitting.py:20-165
# ----------------------------------------------------------------------------------------------


def get_seeds(im, max_num_seeds=None, th_seed=150, th_seed_per=95, use_percentile=False,
              sel_center=None, seed_radius=30, gfilt_size=0.75, background_gfilt_size=7.5,
              filt_size=3, min_edge_distance=2, use_dynamic_th=True, dynamic_niters=10,
              min_dynamic_seeds=1, remove_hot_pixel=True, hot_pixel_th=3, return_h=False,
              return_th=False):
    """fitting.py:20-154.  Returns (N,3) or (N,4) float64 [z,x,y(,h)], brightest first."""
    if not isinstance(im, np.ndarray):
        raise TypeError("image given should be a numpy.ndarray")
    if int(filt_size) != 3:
        raise NotImplementedError("oracle restates filt_size=3 only")
    if th_seed_per >= 100 or th_seed_per <= 50:
        use_percentile = False
    if sel_center is not None:                                                    # :56-68
        if len(sel_center) != im.ndim:
            raise IndexError("num of dimensions should match for selected center and image given.")
        center = np.array(sel_center, dtype=int)
        llims = np.max([np.zeros(im.ndim), center - seed_radius], axis=0)
        rlims = np.min([np.array(im.shape), center + seed_radius], axis=0)
        lims = np.array(np.transpose(np.stack([llims, rlims])), dtype=int)
        _im = im[tuple(slice(l, r) for l, r in lims)]
        local_edges = llims
    else:
        local_edges = np.zeros(im.ndim)
        _im = im
    if use_percentile:                                                             # :75-76
        from scipy.stats import scoreatpercentile
        th = scoreatpercentile(im, th_seed_per) - scoreatpercentile(im, (100 - th_seed_per) / 2)
    else:
        th = th_seed
    niters = int(dynamic_niters) if use_dynamic_th else 1
    max_im = gaussian_filter(_im, gfilt_size) if gfilt_size else np.array(_im)     # :91-94
    max_ft = maximum_filter3(max_im) == max_im                                     # :95
    min_im = gaussian_filter(_im, background_gfilt_size) if background_gfilt_size else np.array(_im)
    min_ft = minimum_filter3(min_im) != min_im                                     # :102
    mask = max_ft & min_ft
    diff = max_im.astype(np.float32) - min_im.astype(np.float32)                   # :106
    size = np.array(_im.shape)
    for it in range(niters):                                                       # :113-125
        cur = th * (1 - it / niters)
        coords = np.where(mask & (diff >= cur))
        if min_edge_distance > 0:                                                  # :156-165
            c = np.array(coords).T
            keep = ((c >= min_edge_distance) & (c <= size - min_edge_distance)).all(1) \
                if len(c) else np.zeros(0, dtype=bool)
            coords = tuple(cs[keep] for cs in coords)
        if len(coords[0]) >= min_dynamic_seeds:
            break
    if remove_hot_pixel and len(coords[0]):                                        # :131-138
        xy = coords[1].astype(np.int64) * int(size[2] + 1) + coords[2].astype(np.int64)
        uniq, inv, cts = np.unique(xy, return_inverse=True, return_counts=True)
        keep = cts[inv] < hot_pixel_th
        coords = tuple(cs[keep] for cs in coords)
    hs = diff[coords]                                                              # :140
    final = np.array(coords) + local_edges[:, None]
    if return_h:
        final = np.concatenate([final, hs[None, :]])
    final = np.transpose(final)[np.flipud(np.argsort(hs))]                         # :145
    if max_num_seeds is not None and 0 < max_num_seeds <= len(final):              # :149-150
        final = final[:int(max_num_seeds)]
    if return_th:
        return final, cur
    return final


# ----------------------------------------------------------------------------------------------
# (a3) External/Fitting_v4.py:165-396 — model, Jacobian, LM
# ----------------------------------------------------------------------------------------------

_LOGMAX64 = np.log(np.finfo(np.float64).max)


def _sig_center(c_, delta, c0):
    """Fitting_v4.py:189-217 (one axis)."""
    lim = np.log(np.finfo(np.asarray(c_).dtype).max)
    if c_ >= lim:
        return -delta + c0
    if c_ <= -lim:
        return delta + c0
    return 2. * delta / (1. + np.exp(c_)) - delta + c0


def _sig_sine(t_):
    """Fitting_v4.py:219-229."""
    lim = np.log(np.finfo(np.asarray(t_).dtype).max)
    if t_ >= lim:
        return -1
    if t_ <= -lim:
        return 1
    return 2. / (1 + np.exp(t_)) - 1.


def _sig_ws(w_, min_ws, max_ws):
    """Fitting_v4.py:231-242 (min_ws/max_ws are squared widths)."""
    lim = np.log(np.finfo(np.asarray(w_).dtype).max)
    if w_ >= lim:
        return min_ws
    if w_ <= -lim:
        return (max_ws - min_ws) + min_ws
    return (max_ws - min_ws) / (1. + np.exp(w_)) + min_ws


def _norm_w(w, minw, maxw):
    """Fitting_v4.py:369-375."""
    if w > 0:
        e = np.exp(-w)
        return 0.5 * (maxw - minw) * e / (maxw * e + minw) ** 2
    e = np.exp(w)
    return 0.5 * (maxw - minw) * e / (minw * e + maxw) ** 2


class GaussianFit(object):
    """Restatement of Fitting_v4.GaussianFit (:165-396).  ``X`` rows are (z,x,y) = "x,y,z" of
    the reference's internal naming; p = [h, c0, c1, c2, bk, w0, w1, w2, sin_t, sin_p, eps]."""

    def __init__(self, im, X, center=None, n_aprox=10, min_w=0.5, max_w=4., delta_center=3.,
                 init_w=1.5):
        self.min_w = min_w * min_w
        self.max_w = max_w * max_w
        self.delta_center = delta_center
        self.im = np.array(im, dtype=np.float32)
        self.x, self.y, self.z = np.array(X, dtype=np.float32)
        argsort_im = np.argsort(im)
        if center is None:
            center = np.median(X[:, argsort_im][:, -n_aprox:], -1)
        self.center_est = center
        sorted_im = im[argsort_im]
        eps = np.exp(-10.)
        bk_guess = np.log(np.max([np.mean(sorted_im[:n_aprox]), eps]))
        h_guess = np.log(np.max([np.mean(sorted_im[-n_aprox:]), eps]))
        wsq = init_w ** 2
        wg = np.log((self.max_w - wsq) / (wsq - self.min_w))
        self.p_ = np.array([bk_guess, h_guess, 0, 0, 0, wg, wg, wg, 0, 0], dtype=np.float32)
        self.to_natural_paramaters()
        self.success = False

    def _geom(self, parms):
        bk, h, xp, yp, zp, w1, w2, w3, pp, tp = parms
        t, p = _sig_sine(tp), _sig_sine(pp)
        ws1, ws2, ws3 = (_sig_ws(w, self.min_w, self.max_w) for w in (w1, w2, w3))
        d = self.delta_center
        xc = _sig_center(xp, d, self.center_est[0])
        yc = _sig_center(yp, d, self.center_est[1])
        zc = _sig_center(zp, d, self.center_est[2])
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
        return (bk, h, xp, yp, zp, w1, w2, w3, pp, tp, t, p, tc, pc, t2, p2, tc2, pc2,
                s1, s2, s3, x2c, y2c, z2c, xyc, xzc, yzc, xt, yt, zt)

    def calc_f(self, parms):                                                       # :259-290
        self.p_ = parms
        g = self._geom(parms)
        bk, h = g[0], g[1]
        x2c, y2c, z2c, xyc, xzc, yzc, xt, yt, zt = g[21:30]
        xsigmax = x2c * xt * xt + y2c * yt * yt + z2c * zt * zt + xyc * xt * yt + xzc * xt * zt + yzc * yt * zt
        self.f0 = np.exp(h - 0.5 * xsigmax)
        bk = np.clip(bk, -709.78, 709.78)
        self.f = np.exp(bk) + self.f0
        return self.f

    def calc_eps(self, parms):                                                     # :295-320
        return self.calc_f(parms) - self.im

    def calc_jac(self, parms):                                                     # :321-367
        (bk, h, xp, yp, zp, w1, w2, w3, pp, tp, t, p, tc, pc, t2, p2, tc2, pc2,
         s1, s2, s3, x2c, y2c, z2c, xyc, xzc, yzc, xt, yt, zt) = self._geom(parms)
        xt2, xtyt, xtzt, yt2, ytzt, zt2 = xt * xt, xt * yt, xt * zt, yt * yt, yt * zt, zt * zt
        xsigmax = x2c * xt2 + y2c * yt2 + z2c * zt2 + xyc * xtyt + xzc * xtzt + yzc * ytzt
        d, minw, maxw = self.delta_center, self.min_w, self.max_w
        f2 = np.exp(h - 0.5 * xsigmax)
        f1 = np.exp(bk) + np.zeros(len(f2))
        e_xp, e_yp, e_zp = np.exp(-np.abs(xp)), np.exp(-np.abs(yp)), np.exp(-np.abs(zp))
        norm_xp = -d * e_xp / ((1 + e_xp) * (1 + e_xp))
        norm_yp = -d * e_yp / ((1 + e_yp) * (1 + e_yp))
        norm_zp = -d * e_zp / ((1 + e_zp) * (1 + e_zp))
        f3 = (f2 * (2 * x2c * xt + xyc * yt + xzc * zt)) * norm_xp
        f4 = (f2 * (xt * xyc + 2 * y2c * yt + yzc * zt)) * norm_yp
        f5 = (f2 * (xt * xzc + yt * yzc + 2 * z2c * zt)) * norm_zp
        f6 = (f2 * (-pc2 * tc2 * xt2 - 2 * pc2 * t * tc * xtyt - pc2 * t2 * yt2 + 2 * p * pc * tc * xtzt
                    + 2 * p * pc * t * ytzt - p2 * zt2)) * _norm_w(w1, minw, maxw)
        f7 = (f2 * (-t2 * xt2 + 2 * t * tc * xtyt - tc2 * yt2)) * _norm_w(w2, minw, maxw)
        f8 = (f2 * (-p2 * tc2 * xt2 - 2 * p2 * t * tc * xtyt - p2 * t2 * yt2 - 2 * p * pc * tc * xtzt
                    - 2 * p * pc * t * ytzt - pc2 * zt2)) * _norm_w(w3, minw, maxw)
        e_p = np.exp(-np.abs(pp) / 2)
        norm_p = e_p / (1 + e_p * e_p)
        f9 = f2 * (s3 - s1) * ((2 * pc2 - 1.) * (tc * xtzt + t * ytzt)
                               + p * pc * (tc2 * xt2 + 2 * t * tc * xtyt + t2 * yt2 - zt2)) * norm_p
        e_t = np.exp(-np.abs(tp) / 2)
        norm_t = e_t / (1 + e_t * e_t)
        f10 = f2 * ((pc2 * s1 - s2 + p2 * s3) * (t * tc * (yt2 - xt2) - (t2 - tc2) * xtyt)
                    + p * pc * (s1 - s3) * (t * xtzt - tc * ytzt)) * norm_t
        self.jac = np.array([f1, f2, f3, f4, f5, f6, f7, f8, f9, f10], np.float32).T
        return self.jac

    def to_natural_paramaters(self, parms=None):                                   # :244-258
        if parms is None:
            parms = self.p_
        bk, h, xp, yp, zp, w1, w2, w3, pp, tp = parms
        bkf, hf = np.exp(bk), np.exp(h)
        t, p = _sig_sine(tp), _sig_sine(pp)
        w1f, w2f, w3f = (np.sqrt(_sig_ws(w, self.min_w, self.max_w)) for w in (w1, w2, w3))
        d = self.delta_center
        xc = _sig_center(xp, d, self.center_est[0])
        yc = _sig_center(yp, d, self.center_est[1])
        zc = _sig_center(zp, d, self.center_est[2])
        eps = np.mean(np.abs(self.calc_eps(parms)))
        self.p = np.array([hf, xc, yc, zc, bkf, w1f, w2f, w3f, t, p, eps], dtype=np.float32)
        return self.p

    def fit(self):                                                                 # :377-393
        from scipy.optimize import leastsq
        if len(self.p_) > len(self.im):
            self.success = False
        else:
            parmsf, _, info, _, self.ier = leastsq(self.calc_eps, self.p_, Dfun=self.calc_jac, maxfev=1000,
                                                    full_output=True)
            self.nfev = int(info["nfev"])   # bookkeeping for the tests: a fit that stops at maxfev (ier 5) is unconverged
            self.p_ = parmsf
            self.to_natural_paramaters()
            self.center = self.p[1:4]
            self.success = True

    def get_im(self):                                                              # :394-396
        self.calc_f(self.p_)
        return self.f0


# ----------------------------------------------------------------------------------------------
# (a4) External/Fitting_v4.py:559-683
# ----------------------------------------------------------------------------------------------


def ball_offsets(radius):
    """Fitting_v4.py:580-583 — offsets in [-r, r)^3 with d^2 <= r^2 (512 for r=5), C order."""
    zb, xb, yb = np.reshape(np.indices([radius * 2] * 3) - radius, [3, -1])
    keep = zb * zb + xb * xb + yb * yb <= radius ** 2
    return zb[keep], xb[keep], yb[keep]


def _in_dim(x, y, z, xmax, ymax, zmax):                                            # :399-401
    keep = (x >= 0) & (x < xmax) & (y >= 0) & (y < ymax) & (z >= 0) & (z < zmax)
    return x[keep], y[keep], z[keep]


class iter_fit_seed_points(object):
    """Restatement of Fitting_v4.iter_fit_seed_points.

    ``voronoi='ckdtree'`` follows the reference (cKDTree.query; tie winner tree-dependent),
    ``voronoi='lowest_index'`` is the deterministic rule the HIP kernel implements (a voxel
    equidistant from two seeds goes to the lower seed index).  They differ only for voxels
    exactly equidistant from two seeds closer than 2r.
    """

    def __init__(self, im, centers, radius_fit=5, min_delta_center=1., max_delta_center=2.5,
                 n_max_iter=10, max_dist_th=0.1, min_w=0.5, max_w=4, init_w=1.5,
                 voronoi="ckdtree"):
        self.im = im
        self.radius_fit = radius_fit
        self.n_max_iter = n_max_iter
        self.max_dist_th = max_dist_th
        self.min_delta_center = min_delta_center
        self.max_delta_center = max_delta_center
        self.centers = centers.T
        self.zb, self.xb, self.yb = ball_offsets(radius_fit)
        self.sz, self.sx, self.sy = im.shape
        self.min_w, self.max_w, self.init_w = min_w, max_w, init_w
        self.voronoi = voronoi

    def _gfit(self, im_, X, center, delta_center):
        return GaussianFit(im_, X, center=center, delta_center=delta_center,
                           min_w=self.min_w, max_w=self.max_w, init_w=self.init_w)

    def _nearest_is_me(self, X_full, ic):
        pts = X_full.T
        if self.voronoi == "ckdtree":
            _, nn = self.tree.query(pts, distance_upper_bound=self.radius_fit * 2)
            return nn == ic
        c = np.asarray(self.centers, dtype=np.float64)
        d2 = ((c - c[ic]) ** 2).sum(1)
        near = np.where(d2 <= (2.0 * self.radius_fit) ** 2)[0]  # a voxel 5 away from two seeds 10 apart is a tie
        dv = ((pts[:, None, :].astype(np.float64) - c[near][None]) ** 2).sum(-1)
        mine = dv[:, list(near).index(ic)]
        keep = np.ones(len(pts), dtype=bool)
        for k, j in enumerate(near):
            if j == ic:
                continue
            keep &= ~((dv[:, k] < mine) | ((dv[:, k] == mine) & (j < ic)))
        return keep

    def firstfit(self):                                                            # :590-639
        if len(self.centers) > 0:
            from scipy.spatial import cKDTree
            self.ps, self.ims_rec, self.centers_fit, self.success, self.gparms = [], [], [], [], []
            self.im_subtr = np.array(self.im, dtype=float)
            self.nfev_last = np.zeros(len(self.centers), dtype=int)   # evaluations of each seed's latest fit
            self.nfev_peak = np.zeros(len(self.centers), dtype=int)   # ... and of its longest fit so far
            self.tree = cKDTree(self.centers)
            for ic, (zc, xc, yc) in enumerate(self.centers):
                z, x, y = int(zc) + self.zb, int(xc) + self.xb, int(yc) + self.yb
                z, x, y = _in_dim(z, x, y, self.sz, self.sx, self.sy)
                X_full = np.array([z, x, y], dtype=int)
                keep = self._nearest_is_me(X_full, ic)
                X = X_full[:, keep]
                im_ = self.im[X[0], X[1], X[2]]
                obj = self._gfit(im_, X, [zc, xc, yc], self.min_delta_center)
                obj.fit()
                self.nfev_last[ic] = getattr(obj, "nfev", 0)
                self.nfev_peak[ic] = max(self.nfev_peak[ic], self.nfev_last[ic])
                self.gparms.append([im_, X, [zc, xc, yc]])
                self.success.append(obj.success)
                if obj.success:
                    self.ps.append(obj.p)
                    self.centers_fit.append(obj.center)
                    obj.x, obj.y, obj.z = X_full
                    im_rec = obj.get_im()
                    self.ims_rec.append(im_rec)
                    self.im_subtr[X_full[0], X_full[1], X_full[2]] -= im_rec
                else:
                    self.ims_rec.append(np.nan)
                    self.ps.append([np.nan] * len(obj.p))
                    self.centers_fit.append([np.nan] * 3)
        self.im_add = np.array(self.im_subtr)

    def repeatfit(self):                                                           # :641-683
        self.n_iter = 0
        self.converged = np.zeros(len(self.centers), dtype=bool)
        self.dists = np.zeros(len(self.centers)) + np.inf
        converged = np.all(self.converged)
        while not converged:
            self.success_old, self.centers_fit_old = np.array(self.success), np.array(self.centers_fit)
            for ic, (zc, xc, yc) in enumerate(self.centers):
                if not self.converged[ic]:
                    z, x, y = int(zc) + self.zb, int(xc) + self.xb, int(yc) + self.yb
                    z, x, y = _in_dim(z, x, y, self.sz, self.sx, self.sy)
                    X = np.array([z, x, y])
                    im_ = self.im_add[z, x, y]
                    if self.success_old[ic]:
                        im_ = self.ims_rec[ic] + im_
                    obj = self._gfit(im_, X, [zc, xc, yc], self.max_delta_center)
                    obj.fit()
                    self.nfev_last[ic] = getattr(obj, "nfev", 0)
                    self.nfev_peak[ic] = max(self.nfev_peak[ic], self.nfev_last[ic])
                    self.success[ic] = obj.success
                    if obj.success:
                        im_rec = obj.get_im()
                        self.ps[ic] = obj.p
                        self.centers_fit[ic] = obj.center
                        self.ims_rec[ic] = im_rec
                        self.im_add[z, x, y] = im_ - im_rec
            keep = (np.array(self.success) & np.array(self.success_old)) > 0
            self.dists[~keep] = 0
            self.dists[keep] = np.sum((np.array(self.centers_fit_old)[keep]
                                       - np.array(self.centers_fit)[keep]) ** 2, axis=-1)
            self.converged = self.dists < self.max_dist_th ** 2
            converged = np.all(self.converged)
            self.n_iter += 1
            converged = converged or (self.n_iter > self.n_max_iter)


# ----------------------------------------------------------------------------------------------
# (a2) spot_tools/fitting.py:169-262 ; (a5) :268-363
# ----------------------------------------------------------------------------------------------


def find_image_background(im, dtype=np.uint16, bin_size=10, max_iter=10):
    """io_tools/load.py:642-687 — histogram-peak background: highest peak found by
    scipy.signal.find_peaks with a height threshold halved from size/50 until a peak exists."""
    from scipy.signal import find_peaks
    if dtype is None:
        dtype = im.dtype
    cts, bins = np.histogram(im, bins=np.arange(np.iinfo(dtype).min, np.iinfo(dtype).max, bin_size))
    peaks = []
    height = np.size(im) / 50
    it = 0
    while len(peaks) == 0:
        height = height / 2
        peaks, params = find_peaks(cts, height=height)
        it += 1
        if it > max_iter:
            break
    if it > max_iter:
        return np.nanmedian(im)
    sel = peaks[np.argmax(params['peak_heights'])]
    return (bins[sel] + bins[sel + 1]) / 2


def fit_fov_image(im, channel=None, seeds=None, seed_mask=None, max_num_seeds=500, th_seed=300,
                  th_seed_per=95, use_percentile=False, use_dynamic_th=True, dynamic_niters=10,
                  min_dynamic_seeds=1, remove_hot_pixel=True, seeding_kwargs={}, fit_radius=5,
                  normalize_background=False, normalize_local=False, background_args={},
                  fitting_args={}, remove_boundary_points=True, voronoi="ckdtree",
                  return_fitter=False):
    """fitting.py:169-262."""
    th_seed = float(th_seed)
    if seeds is None:
        _seeds = get_seeds(im, max_num_seeds=max_num_seeds, th_seed=th_seed, th_seed_per=th_seed_per,
                           use_percentile=use_percentile, use_dynamic_th=use_dynamic_th,
                           dynamic_niters=dynamic_niters, min_dynamic_seeds=min_dynamic_seeds,
                           remove_hot_pixel=remove_hot_pixel, return_h=False, **seeding_kwargs)
    else:
        _seeds = np.array(seeds)[:, :im.ndim]
    if len(_seeds) == 0:
        return np.array([])
    if seed_mask is not None:
        sel = [s for s in _seeds if seed_mask[tuple(np.round(s[:im.ndim]).astype(np.int32))] > 0]
        _seeds = np.array(sel)
    fitter = iter_fit_seed_points(im, _seeds.T, radius_fit=fit_radius, voronoi=voronoi, **fitting_args)
    fitter.firstfit()
    fitter.repeatfit()
    spots = np.array(fitter.ps)
    spots = spots[np.sum(np.isnan(spots), axis=1) == 0]
    if remove_boundary_points:
        kept = (spots[:, 1:4] > np.zeros(3)).all(1) * (spots[:, 1:4] < np.array(im.shape)).all(1)
        spots = spots[np.where(kept)[0]]
    if normalize_background and not normalize_local:
        spots[:, 0] = spots[:, 0] / find_image_background(im, **background_args)
    elif normalize_local:
        backs = []
        for pt in spots:
            crop = neighboring_crop(pt[1:4], fit_radius * 2, np.array(im.shape))
            backs.append(find_image_background(im[crop], **background_args))
        spots[:, 0] = spots[:, 0] / np.array(backs)
    if return_fitter:
        return spots, fitter
    return spots


def neighboring_crop(coord, crop_size, single_im_size):
    """io_tools/crop.py:59-88 generate_neighboring_crop → tuple of slices."""
    coord = np.array(coord)[:len(single_im_size)]
    left = np.max([np.round(coord - crop_size), np.zeros(len(coord))], axis=0).astype(np.int32)
    right = np.min([np.round(coord + crop_size + 1), np.array(single_im_size)], axis=0).astype(np.int32)
    return tuple(slice(int(l), int(r)) for l, r in zip(left, right))


def select_sparse_centers(centers, distance_th=9, distance_norm=np.inf):
    """fitting.py:338-363 — greedy selection, keep a centre if no kept centre is within th."""
    sel = []
    for ct in centers:
        if len(sel) == 0:
            sel.append(ct)
        else:
            d = np.linalg.norm(np.array(sel) - ct[np.newaxis, :], axis=1, ord=distance_norm)
            if not (d <= distance_th).any():
                sel.append(ct)
    return np.array(sel)


def get_centers(im, seeds=None, th_seed=150, th_seed_per=98, use_percentile=False, sel_center=None,
                seed_radius=40, max_num_seeds=None, use_dynamic_th=True, min_num_seeds=1,
                remove_hot_pixel=True, hot_pixel_th=3, seed_kwargs={}, fit_radius=5,
                remove_close_pts=True, close_threshold=0.1, voronoi="ckdtree"):
    """fitting.py:268-334."""
    if seeds is None:
        seeds = get_seeds(im, max_num_seeds=max_num_seeds, th_seed=th_seed, th_seed_per=th_seed_per,
                          use_percentile=use_percentile, sel_center=sel_center, seed_radius=seed_radius,
                          use_dynamic_th=use_dynamic_th, min_dynamic_seeds=min_num_seeds,
                          remove_hot_pixel=remove_hot_pixel, hot_pixel_th=hot_pixel_th, return_h=False,
                          **seed_kwargs)
    fitter = iter_fit_seed_points(im, seeds.T, radius_fit=fit_radius, voronoi=voronoi)
    fitter.firstfit()
    fitter.repeatfit()
    pfits = fitter.ps
    if len(pfits) > 0:
        centers = np.array(pfits)[:, 1:4]
        if remove_close_pts:
            remove = np.zeros(len(centers), dtype=bool)
            for i, bead in enumerate(centers):
                if np.isnan(bead).any() or np.sum(np.sum((centers - bead) ** 2, axis=1) < close_threshold) > 1:
                    remove[i] = True
                if (bead < 0).any() or (bead > np.array(im.shape)).any():
                    remove[i] = True
            centers = centers[remove == False]  # noqa: E712
    else:
        centers = np.array([])
    return centers


# ----------------------------------------------------------------------------------------------
# (a8) correction_tools/alignment.py:80-135
# ----------------------------------------------------------------------------------------------


def generate_drift_crops(single_im_size, coord_sel=None, drift_size=None):
    """alignment.py:87-135 — 8 crops (8,3,2) int around quarter/half anchor points."""
    size = np.array(single_im_size)
    if coord_sel is None:
        coord_sel = np.array(size / 2, dtype=int)
    if coord_sel[-2] >= size[-2] or coord_sel[-1] >= size[-1]:
        raise ValueError("wrong input coord_sel")
    if drift_size is None:
        drift_size = int(np.max(size) / 4)
    cz, cx, cy = coord_sel[-3] / 2, coord_sel[-2], coord_sel[-1]
    sx, sy = size[-2], size[-1]
    cts = [(cz, cx / 2, cy / 2), (cz, (cx + sx) / 2, (cy + sy) / 2), (cz, (cx + sx) / 2, cy / 2),
           (cz, cx / 2, (cy + sy) / 2), (cz, cx, cy / 2), (cz, cx, (cy + sy) / 2),
           (cz, cx / 2, cy), (cz, (cx + sx) / 2, cy)]
    r = drift_size / 2
    crops = [[[max(c - r, 0), min(c + r, s)] for c, s in zip(ct, size)] for ct in cts]  # :80-85
    return np.array(crops, dtype=int)


# ----------------------------------------------------------------------------------------------
# (a10) alignment_tools.py:286-353 ; spot_tools/matching.py:148-287 ; alignment.py:139-216
# ----------------------------------------------------------------------------------------------


def fftalign_2d(im1, im2, center=(0, 0), max_disp=150):
    """alignment_tools.py:286-328 — normalised full cross-correlation, argmax in ±max_disp."""
    from scipy.signal import fftconvolve
    im2_ = np.array(im2[::-1, ::-1], dtype=float)
    im2_ -= np.mean(im2_)
    im2_ /= np.std(im2_)
    im1_ = np.array(im1, dtype=float)
    im1_ -= np.mean(im1_)
    im1_ /= np.std(im1_)
    cor = fftconvolve(im1_, im2_, mode="full")
    sx, sy = cor.shape
    c = np.array(center) + np.array([sx, sy]) / 2.
    x_min = int(min(max(c[0] - max_disp, 0), sx))
    x_max = int(min(max(c[0] + max_disp, 0), sx))
    y_min = int(min(max(c[1] - max_disp, 0), sy))
    y_max = int(min(max(c[1] + max_disp, 0), sy))
    win = np.zeros_like(cor)
    win[x_min:x_max, y_min:y_max] = 1
    cor = cor * win
    y, x = np.unravel_index(np.argmax(cor), cor.shape)
    xt, yt = (-np.floor(np.array(cor.shape) / 2) + [y, x]).astype(int)
    return xt, yt


def fft3d_from2d(im1, im2, gb=0, max_disp=150):
    """alignment_tools.py:330-353 with gb<=1 (no cv2 blur; the production default, alignment.py:141)."""
    if gb > 1:
        raise NotImplementedError("cv2.blur normalisation is off the default path")
    im1_, im2_ = np.max(im1, 0), np.max(im2, 0)
    tx, ty = fftalign_2d(im1_, im2_, max_disp=max_disp)
    sx, sy = im1_.shape
    im1_t = np.max(im1[:, max(tx, 0):sx + tx, max(ty, 0):sy + ty], axis=-1)
    im2_t = np.max(im2[:, max(-tx, 0):sx - tx, max(-ty, 0):sy - ty], axis=-1)
    tz, _ = fftalign_2d(im1_t, im2_t, max_disp=max_disp)
    return np.array([tz, tx, ty])


def find_paired_centers(tar_cts, ref_cts, drift=None, cutoff=2, dimension=3):
    """matching.py:148-222 — unique nearest pairing within cutoff after shifting ref by drift."""
    from scipy.spatial.distance import cdist
    tar, ref = np.array(tar_cts), np.array(ref_cts)
    if tar.shape[1] > 3:
        tar = tar[:, 1:1 + dimension]
    if ref.shape[1] > 3:
        ref = ref[:, 1:1 + dimension]
    drift = np.zeros(tar.shape[1]) if drift is None else np.array(drift, dtype=float)[:dimension]
    dists = cdist(tar, ref + drift)
    ti, ri = np.where(dists <= cutoff)
    ut = np.where(np.sum(dists <= cutoff, axis=1) == 1)[0]
    ur = np.where(np.sum(dists <= cutoff, axis=0) == 1)[0]
    pairs = [[t, r] for t, r in zip(ti, ri) if t in ut and r in ur]
    ptar = np.array([tar[t] for t, _ in pairs])
    pref = np.array([ref[r] for _, r in pairs])
    return np.nanmean(ptar - pref, axis=0), ptar, pref


def check_paired_centers(paired_tar_cts, paired_ref_cts, outlier_sigma=1.5):
    """matching.py:224-287 — Delaunay-neighbour inverse-distance-weighted outlier rejection."""
    from scipy.spatial import Delaunay
    tar = np.array(paired_tar_cts, dtype=float)
    ref = np.array(paired_ref_cts, dtype=float)
    shifts = tar - ref
    tri = Delaunay(ref)
    new_shifts = []
    simplices = tri.simplices.copy()
    for i, (s, tc, rc) in enumerate(zip(shifts, tar, ref)):
        nb = np.unique(simplices[(simplices == i).any(1)])
        nb = nb[(nb != i) & (nb != -1)]
        w = 1 / np.linalg.norm(ref[nb] - rc, axis=1)
        new_shifts.append(np.dot(shifts[nb].T, w) / np.sum(w))
    new_shifts = np.array(new_shifts)
    diffs = np.linalg.norm(new_shifts - shifts, axis=1)
    keep = np.array(diffs < np.mean(diffs) + np.std(diffs) * outlier_sigma)
    return np.nanmean(tar[keep] - ref[keep], axis=0), tar[keep], ref[keep]


def align_beads(tar_cts, ref_cts, tar_im, ref_im, fft_filt_size=0, match_distance_th=2.,
                check_paired_cts=True, outlier_sigma=1.5):
    """alignment.py:139-216 (use_fft=True branch)."""
    tar_cts, ref_cts = np.array(tar_cts), np.array(ref_cts)
    if np.shape(tar_im) != np.shape(ref_im):
        raise IndexError("tar_im shape should match ref_im shape")
    rough = fft3d_from2d(tar_im, ref_im, gb=fft_filt_size, max_disp=np.max(np.shape(tar_im)) / 2)
    drift, ptar, pref = find_paired_centers(tar_cts, ref_cts, rough, cutoff=float(match_distance_th))
    if check_paired_cts and len(pref) > 3:
        drift, ptar, pref = check_paired_centers(ptar, pref, outlier_sigma=outlier_sigma)
    return drift, ptar, pref


# ----------------------------------------------------------------------------------------------
# phase cross-correlation — PARITY UNPINNED (scikit-image absent; published algorithm:
# Guizar-Sicairos, Thurman & Fienup, Opt. Lett. 33, 156 (2008)); call sites alignment.py:491-494,
# 631-632, classes/preprocess.py:831-835.
# ----------------------------------------------------------------------------------------------


def _upsampled_dft(data, region, upsample, offsets):
    """Matrix-multiply DFT of ``data`` on a ``region``-wide grid upsampled ``upsample``x.

    Axes are contracted last-to-first; each tensordot moves the new axis to the front, so after
    ndim steps the axis order is restored."""
    dims = list(zip(data.shape, offsets))
    for n, off in dims[::-1]:
        k = (np.arange(region) - off)[:, None] * np.fft.fftfreq(n, upsample)
        kern = np.exp(-2j * np.pi * k)
        data = np.tensordot(kern, data, axes=(1, -1))
    return data


def phase_cross_correlation(reference_image, moving_image, upsample_factor=1, normalization="phase"):
    """Restatement of the published efficient sub-pixel registration algorithm.

    Returns (shift, error, phasediff); ``shift`` is what must be applied to ``moving_image``
    to register it onto ``reference_image`` (ref - src), the quantity ``align_image`` consumes.
    ``error``/``phasediff`` are informational (align_image discards them, alignment.py:631).
    """
    src = np.fft.fftn(np.asarray(reference_image, dtype=np.float64))
    tgt = np.fft.fftn(np.asarray(moving_image, dtype=np.float64))
    shape = src.shape
    prod = src * tgt.conj()
    if normalization == "phase":
        eps = np.finfo(prod.real.dtype).eps
        prod /= np.maximum(np.abs(prod), 100 * eps)
    elif normalization is not None:
        raise ValueError("normalization must be 'phase' or None")
    cc = np.fft.ifftn(prod)
    maxima = np.unravel_index(np.argmax(np.abs(cc)), cc.shape)
    mid = np.array([np.fix(s / 2) for s in shape])
    shift = np.stack(maxima).astype(np.float64)
    shift[shift > mid] -= np.array(shape)[shift > mid]
    if upsample_factor == 1:
        s_amp = np.sum(np.real(src * src.conj())) / src.size
        t_amp = np.sum(np.real(tgt * tgt.conj())) / tgt.size
        ccmax = cc[maxima]
    else:
        uf = float(upsample_factor)
        shift = np.round(shift * uf) / uf
        region = int(np.ceil(uf * 1.5))
        dftshift = np.fix(region / 2.0)
        offs = dftshift - shift * uf
        cc_up = _upsampled_dft(prod.conj(), region, uf, offs).conj()
        maxima = np.unravel_index(np.argmax(np.abs(cc_up)), cc_up.shape)
        ccmax = cc_up[maxima]
        shift = shift + (np.stack(maxima).astype(np.float64) - dftshift) / uf
        s_amp = np.sum(np.real(src * src.conj()))
        t_amp = np.sum(np.real(tgt * tgt.conj()))
    for d in range(len(shape)):
        if shape[d] == 1:
            shift[d] = 0
    amp = s_amp * t_amp
    err = np.sqrt(np.abs(1.0 - (ccmax * ccmax.conj()).real / amp)) if amp != 0 else np.nan
    return shift, float(err), float(np.arctan2(ccmax.imag, ccmax.real))


# ----------------------------------------------------------------------------------------------
# (a9) correction_tools/alignment.py:527-695
# ----------------------------------------------------------------------------------------------

_default_align_fitting_args = {'th_seed': 300, 'th_seed_per': 95, 'use_percentile': False,
                               'use_dynamic_th': True, 'min_dynamic_seeds': 10, 'max_num_seeds': 200}


def consensus_drift(drifts_iter, min_good_drifts=3, drift_diff_th=1.):
    """alignment.py:624-695 — consume per-crop drifts lazily; early exit on consensus."""
    drifts = []
    for dft in drifts_iter:
        drifts.append(dft)
        mean = np.nanmean(drifts, axis=0)
        if len(drifts) >= min_good_drifts:
            dists = np.linalg.norm(drifts - mean, axis=1)
            kept = np.where(dists <= drift_diff_th)[0]
            if len(kept) >= min_good_drifts:
                return np.nanmean(np.array(drifts)[kept], axis=0), 0
    from scipy.spatial.distance import pdist, squareform
    drifts = np.array(drifts)
    dm = squareform(pdist(drifts))
    np.fill_diagonal(dm, np.inf)
    sel = np.array(np.unravel_index(np.argmin(dm), dm.shape))
    sel_drifts = list(drifts[sel])
    sel_drifts.append(drifts[np.argmin(dm[:, sel].sum(1))])
    return np.nanmean(sel_drifts, axis=0), 1


def align_image(src_im, ref_im, crop_list=None, use_autocorr=True, precision_fold=100,
                min_good_drifts=3, drift_diff_th=1., fitting_args={}, match_distance_th=2.,
                normalization="phase", voronoi="ckdtree"):
    """alignment.py:527-695 for ndarray inputs."""
    if np.shape(src_im) != np.shape(ref_im):
        raise IndexError("shape of target image and reference image doesnt match!")
    if crop_list is None:
        crop_list = generate_drift_crops(np.shape(src_im))
    for crop in crop_list:
        if np.shape(np.array(crop)) != (3, 2):
            raise IndexError("crop should be 3x2 np.ndarray.")
    fargs = dict(_default_align_fitting_args)
    fargs.update(fitting_args)

    def per_crop():
        for crop in crop_list:
            s = tuple(slice(*np.array(c, dtype=int)) for c in crop)
            sim, rim = src_im[s], ref_im[s]
            if use_autocorr:
                dft, _, _ = phase_cross_correlation(rim, sim, upsample_factor=precision_fold,
                                                    normalization=normalization)
            else:
                ss = fit_fov_image(sim, None, voronoi=voronoi, **fargs)
                sc = select_sparse_centers(ss[:, 1:4], match_distance_th)
                rs = fit_fov_image(rim, None, voronoi=voronoi, **fargs)
                rc = select_sparse_centers(rs[:, 1:4], match_distance_th)
                dft, _, _ = align_beads(sc, rc, sim, rim, match_distance_th=match_distance_th)
                dft = dft * -1
            yield dft

    return consensus_drift(per_crop(), min_good_drifts, drift_diff_th)


# ----------------------------------------------------------------------------------------------
# (a11) correction_tools/translate.py:5-31 and production twins io_tools/load.py:438-453
# ----------------------------------------------------------------------------------------------


def warp_3d_image(image, drift, chromatic_profile=None, warp_order=1, border_mode="constant"):
    """translate.py:5-31 via scipy.ndimage.map_coordinates (third-party spline arithmetic)."""
    from scipy.ndimage import map_coordinates
    size = np.array(image.shape)
    coords = np.meshgrid(np.arange(size[0]), np.arange(size[1]), np.arange(size[2]))
    coords = np.stack(coords).transpose((0, 2, 1, 3))
    if chromatic_profile is not None:
        coords = coords + chromatic_profile
    drift = np.array(drift)
    if drift.any():
        coords = coords - drift[:, None, None, None]
    out = map_coordinates(image, coords.reshape(coords.shape[0], -1), order=warp_order,
                          mode=border_mode, cval=np.min(image))
    return out.reshape(image.shape).astype(image.dtype)


def spline3_mirror_line(c):
    """scipy.ndimage's cubic B-spline prefilter of one line with the whole-sample-symmetric boundary it uses for
    ``mode='constant'`` / ``'mirror'`` (ni_splines.c: _init_causal_mirror, _init_anticausal_mirror; SciPy 1.15.3 in
    this image), operation for operation.  What csrc/warp.hip's spline_mirror_k runs; pinned against
    ``spline_filter1d`` bit for bit in tests/test_host_logic_cpu.py."""
    z = np.float64(-0.26794919243112270647)
    c = np.array(c, dtype=np.float64)
    n = len(c)
    if n < 2:
        return c
    c *= (1.0 - z) * (1.0 - 1.0 / z)
    zn1 = np.power(z, float(n - 1))
    s = c[0] + zn1 * c[n - 1]
    zi = z
    for i in range(1, n - 1):
        s += zi * (c[i] + zn1 * c[n - 1 - i])
        zi *= z
    c[0] = s / (1 - zn1 * zn1)
    for i in range(1, n):
        c[i] += z * c[i - 1]
    c[n - 1] = (z * c[n - 2] + c[n - 1]) * z / (z * z - 1)
    for i in range(n - 2, -1, -1):
        c[i] = z * (c[i + 1] - c[i])
    return c


def cubic_constant_1d(x, coords, cval):
    """``map_coordinates(x, [coords], order=3, mode='constant', cval=cval)`` for a 1-D float64 ``x`` (ni_interpolation.c,
    NI_GeometricTransform): a coordinate below 0 or above n-1 gives cval; taps that leave the array are mirrored about
    its first / last sample; weights and summation order as in the 'nearest' path."""
    n = len(x)
    coef = spline3_mirror_line(x)
    s2 = 2 * n - 2

    def mirror(i):
        if i < 0:
            i = s2 * int(-i / s2) + i
            i = i + s2 if i <= 1 - n else -i
        elif i >= n:
            i -= s2 * int(i / s2)
            if i >= n:
                i = s2 - i
        return i
    out = np.empty(len(coords), dtype=np.float64)
    for q, cc in enumerate(coords):
        if cc < 0 or cc > n - 1:
            out[q] = cval
            continue
        fl = np.floor(cc)
        y = cc - fl
        zz = 1.0 - y
        w1 = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0
        w2 = (zz * zz * (zz - 2.0) * 3.0 + 4.0) / 6.0
        w0 = zz * zz * zz / 6.0
        w = [w0, w1, w2, 1.0 - w0 - w1 - w2]
        t = 0.0
        for k in range(4):
            t = t + coef[mirror(int(fl) - 1 + k)] * w[k]
        out[q] = t
    return out


# ----------------------------------------------------------------------------------------------
# (a13) elementwise stages of correct_fov_image, io_tools/load.py:337-384 (the reference's own NumPy
# expressions, verbatim in structure; corrections.py:479-487 for the z-shift)
# ----------------------------------------------------------------------------------------------


def z_shift_correction(im, dtype=np.uint16):
    """corrections.py:479-487 as called at io_tools/load.py:342 with ``im.astype(np.float32)``."""
    im = im.astype(np.float32)
    nim = im / np.median(im, axis=(1, 2))[:, np.newaxis, np.newaxis] * np.median(im)
    return nim.astype(dtype)


def illumination_correction(im, profile, output_dtype=np.uint16):
    """io_tools/load.py:382."""
    return (im.astype(np.float32) / profile[np.newaxis, :]).astype(output_dtype)


def bleedthrough_correction(ims, bleed_profile, output_dtype=np.uint16):
    """io_tools/load.py:355-367."""
    outs = []
    for i in range(len(ims)):
        nim = np.sum([im * bleed_profile[i, j] for j, im in enumerate(ims)], axis=0)
        nim[nim > np.iinfo(output_dtype).max] = np.iinfo(output_dtype).max
        nim[nim < np.iinfo(output_dtype).min] = np.iinfo(output_dtype).min
        outs.append(nim.astype(output_dtype))
    return outs


# ----------------------------------------------------------------------------------------------
# (a12) legacy per-cell path: visual_tools.py:348-381,1775-1870 ; External/Fitting_v3.py ;
#       classes/__init__.py:57-88
# ----------------------------------------------------------------------------------------------


def legacy_get_seed_points_base(im, gfilt_size=0.75, background_gfilt_size=10, filt_size=3,
                                th_seed=300, hot_pix_th=0, return_h=False):
    """visual_tools.py:348-381 (scipy.ndimage filters restated above; rank filters of any size go through
    scipy.ndimage directly, whose output for max/min is exact)."""
    from scipy.ndimage import maximum_filter, minimum_filter
    max_im = gaussian_filter(im, gfilt_size) if gfilt_size else im
    min_im = gaussian_filter(im, background_gfilt_size) if background_gfilt_size else im
    max_filt = np.array(maximum_filter(max_im, filt_size), dtype=np.int64)
    min_filt = np.array(minimum_filter(min_im, filt_size), dtype=np.int64)
    im_plt2 = (max_filt == max_im) & (min_filt != min_im) & (min_filt != 0)
    z, x, y = np.where(im_plt2)
    keep = (max_filt[z, x, y] - min_filt[z, x, y]) > th_seed
    x, y, z = x[keep], y[keep], z[keep]
    h = max_filt[z, x, y] - min_filt[z, x, y]
    if hot_pix_th > 0:
        xy = x.astype(np.int64) * (int(im.shape[2]) + 1) + y          # same grouping as the str([x,y]) keys
        xy_, inv, cts_ = np.unique(xy, return_inverse=True, return_counts=True)
        keep = ~(cts_ > hot_pix_th)[inv] if len(xy) else np.zeros(0, dtype=bool)
        x, y, z, h = x[keep], y[keep], z[keep], h[keep]
    return np.array([z, x, y, h]) if return_h else np.array([z, x, y])


def legacy_get_seed_in_distance(im, center=None, num_seeds=0, seed_radius=30, gfilt_size=0.75,
                                background_gfilt_size=10, filt_size=3, seed_by_per=False,
                                th_seed_percentile=95, th_seed=300, dynamic=True, dynamic_iters=10,
                                min_dynamic_seeds=2, distance_to_edge=1, hot_pix_th=4, return_h=False):
    """visual_tools.py:1775-1870 (np.float/np.int read as float/int)."""
    from scipy.stats import scoreatpercentile
    from scipy.spatial.distance import cdist
    if center is not None and len(center) != 3:
        raise ValueError('wrong input dimension of center!')
    _dim = np.shape(im)
    _im = im.copy()
    if seed_by_per:
        _im_ints = _im[np.isnan(_im) == False].astype(float)
        _th_seed = scoreatpercentile(_im_ints, th_seed_percentile) - scoreatpercentile(_im_ints, 100 - th_seed_percentile)
    else:
        _th_seed = th_seed
    if center is not None:
        _center = np.array(center, dtype=float)
        _limits = np.zeros([2, 3], dtype=int)
        _limits[0, 1:] = np.array([np.max([x, y]) for x, y in zip(np.zeros(2), _center[1:] - seed_radius)], dtype=int)
        _limits[0, 0] = np.array(np.max([0, _center[0] - seed_radius / 2]), dtype=int)
        _limits[1, 1:] = np.array([np.min([x, y]) for x, y in zip(_dim[1:], _center[1:] + seed_radius)], dtype=int)
        _limits[1, 0] = np.array(np.min([_dim[0], _center[0] + seed_radius / 2]), dtype=int)
        _local_center = _center - _limits[0]
        _cim = _im[_limits[0, 0]:_limits[1, 0], _limits[0, 1]:_limits[1, 1], _limits[0, 2]:_limits[1, 2]]
        if dynamic:
            for _dy_ratio in np.linspace(1, 1 / dynamic_iters, dynamic_iters):
                _cand = legacy_get_seed_points_base(_cim, gfilt_size=gfilt_size,
                                                    background_gfilt_size=background_gfilt_size, filt_size=filt_size,
                                                    th_seed=th_seed * _dy_ratio,
                                                    hot_pix_th=hot_pix_th, return_h=True)
                _distance = cdist(_cand[:3].transpose(), _local_center[np.newaxis, :3]).transpose()[0]
                _seeds = _cand[:, _distance < seed_radius]
                _seeds[:3, :] += _limits[0][:, np.newaxis]
                if num_seeds > 0 and _seeds.shape[1] >= min(num_seeds, min_dynamic_seeds):
                    break
                elif num_seeds == 0 and _seeds.shape[1] >= min_dynamic_seeds:
                    break
        else:
            _seeds = legacy_get_seed_points_base(_cim, gfilt_size=gfilt_size, filt_size=filt_size, th_seed=th_seed,
                                                 hot_pix_th=hot_pix_th, return_h=True)
    else:
        _seeds = legacy_get_seed_points_base(_im, gfilt_size=gfilt_size, filt_size=filt_size, th_seed=_th_seed,
                                             hot_pix_th=hot_pix_th, return_h=True)
    if _seeds.shape[1] > 1:
        _order = np.argsort(_seeds[-1], kind="stable")   # the reference's default sort leaves ties unspecified
        _seeds = _seeds[:, np.flipud(_order[-num_seeds:])]
    return _seeds[:3].transpose() if not return_h else _seeds[:4].transpose()


class GaussianFitV3(GaussianFit):
    """External/Fitting_v3.py:50-262 where it differs from v4: per-axis start widths (:71-79), unguarded sigmoids,
    to_center as written (:81-87), no background clip, leastsq with MINPACK's default maxfev."""

    def __init__(self, im, X, center=None, n_aprox=10, min_w=0.5, max_w=4., delta_center=3.,
                 init_w=(1.35, 1.9, 1.9), weight_sigma=0):
        assert not weight_sigma
        self.min_w = min_w * min_w
        self.max_w = max_w * max_w
        self.delta_center = delta_center
        self.im = np.array(im, dtype=np.float32)
        self.x, self.y, self.z = np.array(X, dtype=np.float32)
        argsort_im = np.argsort(im)
        if center is None:
            center = np.median(X[:, argsort_im][:, -n_aprox:], -1)
        self.center_est = center
        sorted_im = im[argsort_im]
        eps = np.exp(-10.)
        bk_guess = np.log(np.max([np.mean(sorted_im[:n_aprox]), eps]))
        h_guess = np.log(np.max([np.mean(sorted_im[-n_aprox:]), eps]))
        init_w = np.array(init_w[:3]).copy()
        for _i, _iw in enumerate(init_w):
            if _iw ** 2 > max_w or _iw ** 2 < min_w:
                init_w[_i] = 1.5 ** 2
            init_w[_i] = np.log((self.max_w - init_w[_i] ** 2) / (init_w[_i] ** 2 - self.min_w))
        self.p_ = np.array([bk_guess, h_guess, 0, 0, 0, init_w[0], init_w[1], init_w[2], 0, 0], dtype=np.float32)
        self.to_natural_paramaters()
        self.success = False

    def _centers(self, c0_, c1_, c2_):
        d = self.delta_center
        c0 = 2. * d * np.exp(-c0_) / (1. + np.exp(-c0_)) - d + self.center_est[0]
        c1 = 2. * d * np.exp(-c1_) / (1. + np.exp(-c1_)) - d + self.center_est[1]
        c2 = 2. * d * np.exp(-c1_) / (1. + np.exp(-c2_)) - d + self.center_est[2]
        return c0, c1, c2

    def _geom(self, parms):
        g = list(GaussianFit._geom(self, parms))
        xc, yc, zc = self._centers(parms[2], parms[3], parms[4])
        g[27], g[28], g[29] = self.x - xc, self.y - yc, self.z - zc
        return tuple(g)

    def calc_f(self, parms):
        self.p_ = parms
        g = self._geom(parms)
        bk, h = g[0], g[1]
        x2c, y2c, z2c, xyc, xzc, yzc, xt, yt, zt = g[21:30]
        xsigmax = x2c * xt * xt + y2c * yt * yt + z2c * zt * zt + xyc * xt * yt + xzc * xt * zt + yzc * yt * zt
        self.f0 = np.exp(h - 0.5 * xsigmax)
        self.f = np.exp(bk) + self.f0
        return self.f

    def to_natural_paramaters(self, parms=None):
        if parms is None:
            parms = self.p_
        bk, h, xp, yp, zp, w1, w2, w3, pp, tp = parms
        bkf, hf = np.exp(bk), np.exp(h)
        t, p = _sig_sine(tp), _sig_sine(pp)
        w1f, w2f, w3f = (np.sqrt(_sig_ws(w, self.min_w, self.max_w)) for w in (w1, w2, w3))
        xc, yc, zc = self._centers(xp, yp, zp)
        eps = np.mean(np.abs(self.calc_eps(parms)))
        self.p = np.array([hf, xc, yc, zc, bkf, w1f, w2f, w3f, t, p, eps], dtype=np.float32)
        return self.p

    def fit(self):
        from scipy.optimize import leastsq
        if len(self.p_) > len(self.im):
            self.success = False
        else:
            parmsf, _ = leastsq(self.calc_eps, self.p_, Dfun=self.calc_jac)
            self.p_ = parmsf
            self.to_natural_paramaters()
            self.center = self.p[1:4]
            self.success = True


class iter_fit_seed_points_v3(iter_fit_seed_points):
    """External/Fitting_v3.py:312-425: same loop as v4 with GaussianFitV3 and `closest` (cdist + argmin over all
    centres = lowest index on ties) for the first-fit Voronoi cells."""

    def __init__(self, im, centers, radius_fit=5, min_delta_center=1., max_delta_center=2.5, n_max_iter=10,
                 max_dist_th=0.1, init_w=(1.35, 1.9, 1.9), weight_sigma=0):
        iter_fit_seed_points.__init__(self, im, centers, radius_fit, min_delta_center, max_delta_center, n_max_iter,
                                      max_dist_th, voronoi="lowest_index")
        self.init_w3 = init_w

    def _nearest_is_me(self, X_full, ic):
        from scipy.spatial.distance import cdist
        dists = cdist(X_full.T, self.centers)
        center_id = np.argmin(cdist([self.centers[ic]], self.centers)[0, :])
        return np.argmin(dists, axis=-1) == center_id

    def firstfit(self):
        if len(self.centers) == 0:
            raise ValueError(f"{len(self.centers)} points have been seeded, exit.")
        iter_fit_seed_points.firstfit(self)

    def _gfit(self, im_, X, center, delta_center):
        return GaussianFitV3(im_, X, center=center, delta_center=delta_center, init_w=self.init_w3)


def fit_single_image(_im, _id, _chrom_coords, _seeding_args, _fitting_args, _check_fitting=True,
                     _normalization=True):
    """classes/__init__.py:57-88."""
    _spots_for_chrom = []
    if _normalization:
        _norm_cst = np.nanmedian(_im)
    for _chrom_coord in _chrom_coords:
        if _im is None:
            _spots_for_chrom.append(np.array([]))
            continue
        _seeds = legacy_get_seed_in_distance(_im, _chrom_coord, *_seeding_args)
        if len(_seeds) == 0:
            _spots_for_chrom.append(np.array([]))
            continue
        _fitter = iter_fit_seed_points_v3(_im, _seeds.T, *_fitting_args)
        _fitter.firstfit()
        if _check_fitting:
            _fitter.repeatfit()
        _spots = np.array(_fitter.ps)
        if _normalization:
            _spots[:, 0] = _spots[:, 0] / _norm_cst
        _spots_for_chrom.append(_spots)
    return _spots_for_chrom


# ----------------------------------------------------------------------------------------------
# (f1) io_tools/load.py:166-522 correct_fov_image on an in-memory raw movie (profiles passed in)
# ----------------------------------------------------------------------------------------------


def correct_fov_image(raw_im, sel_channels, single_im_size, all_channels, num_buffer_frames=10, num_empty_frames=0,
                      drift=None, drift_channel='488', corr_channels=('750', '647', '561'), hot_pixel_corr=True,
                      hot_pixel_th=4, z_shift_corr=False, illumination_corr=True, illumination_profile=None,
                      bleed_corr=True, bleed_profile=None, chromatic_ref_channel='647', chromatic_corr=True,
                      chromatic_profile=None, gaussian_highpass=False, gauss_sigma=3, gauss_truncate=2,
                      output_dtype=np.uint16, verbose=True):
    """The stage order, dtypes and conditions of the reference function (warp_image=True, no drift calculation,
    no normalisation), composed from the restatements above."""
    sel_channels = [str(c) for c in ([sel_channels] if isinstance(sel_channels, (str, int)) else sel_channels)]
    all_channels = [str(c) for c in all_channels]
    single_im_size = np.array(single_im_size, dtype=int)
    drift = np.zeros(3, np.float32) if drift is None else np.array(drift, dtype=np.float32)
    corr_channels = [str(c) for c in sorted(corr_channels, key=lambda v: -int(v)) if str(c) in all_channels]
    overlap = [c for c in corr_channels if c in sel_channels]
    load_channels = list(corr_channels) if (overlap and bleed_corr) else []
    for c in sel_channels:
        if c not in load_channels:
            load_channels.append(c)
    n_col = int((raw_im.shape[0] - 2 * num_buffer_frames - num_empty_frames) / single_im_size[0])
    chs = all_channels[:n_col]
    starts = [num_empty_frames + num_buffer_frames + (chs.index(c) - num_empty_frames - num_buffer_frames) % n_col
              for c in load_channels]                                                       # :534-548
    ims = [raw_im[s:s + single_im_size[0] * n_col:n_col].copy() for s in starts]
    if hot_pixel_corr:                                                                      # :323-334
        ims = [remove_hot_pixels(im.astype(np.float32), dtype=output_dtype, hot_th=hot_pixel_th) for im in ims]
    if z_shift_corr:                                                                        # :337-345
        ims = [z_shift_correction(im, dtype=output_dtype) for im in ims]
    if overlap and bleed_corr:                                                              # :348-370
        bp = np.array(bleed_profile, dtype=np.float32)
        outs = bleedthrough_correction([ims[load_channels.index(c)] for c in corr_channels], bp, output_dtype)
        for c, o in zip(corr_channels, outs):
            ims[load_channels.index(c)] = o
    if illumination_corr:                                                                   # :373-384
        ims = [illumination_correction(im, illumination_profile[c], output_dtype) for im, c in zip(ims, load_channels)]
    chrom_channels = [c for c in corr_channels if c in sel_channels and c != chromatic_ref_channel]
    for c in sel_channels:                                                                  # :424-453
        if ((chromatic_corr and c in chrom_channels) or drift.any()) and verbose:           # (sic: inside `if verbose`)
            im = ims[load_channels.index(c)]
            prof = chromatic_profile[c] if (chromatic_corr and c in chrom_channels and chromatic_profile[c] is not None) else None
            ims[load_channels.index(c)] = warp_3d_image(im, drift, prof, warp_order=3, border_mode="nearest").astype(output_dtype)
    if gaussian_highpass:                                                                   # :489-498
        ims = [gaussian_high_pass_filter(im, gauss_sigma, gauss_truncate) for im in ims]
    return [ims[load_channels.index(c)].astype(output_dtype).copy() for c in sel_channels]


# ----------------------------------------------------------------------------------------------
# DaxProcesser steps where they differ from correct_fov_image (classes/preprocess.py:464-965)
# ----------------------------------------------------------------------------------------------


def daxp_bleedthrough(ims, correction_pf, image_size, rescale=True):
    """:505-523 — float64 accumulation of `im * pf[i, j]`, min-max rescale to the dtype range, clip, cast."""
    outs = []
    for i in range(len(ims)):
        dtype = ims[i].dtype
        mn, mx = np.iinfo(dtype).min, np.iinfo(dtype).max
        im = np.zeros(image_size)
        for j in range(len(ims)):
            im += ims[j] * correction_pf[i, j]
        if rescale:
            im = (im - np.min(im)) / (np.max(im) - np.min(im)) * mx + mn
        outs.append(np.clip(im, a_min=mn, a_max=mx).astype(dtype))
    return outs


def daxp_illumination(im, pf, rescale=True):
    """:653-662."""
    dtype = im.dtype
    mn, mx = np.iinfo(dtype).min, np.iinfo(dtype).max
    q = im.astype(np.float32) / pf[np.newaxis, :]
    if rescale:
        q = (q - np.min(q)) / (np.max(q) - np.min(q)) * mx + mn
    return np.clip(q, a_min=mn, a_max=mx).astype(dtype)


def daxp_warp(im, drift, chromatic=None):
    """:918-946 — coordinates = (grid - drift) + chromatic, cubic map_coordinates, mode 'nearest'."""
    from scipy.ndimage import map_coordinates
    size = im.shape
    coords = np.meshgrid(np.arange(size[0]), np.arange(size[1]), np.arange(size[2]))
    coords = np.stack(coords).transpose((0, 2, 1, 3))
    drift = np.array(drift)
    if drift.any():
        coords = coords - drift[:, np.newaxis, np.newaxis, np.newaxis]
    if chromatic is not None:
        coords = coords + chromatic
    out = map_coordinates(im, coords.reshape(coords.shape[0], -1), mode='nearest').astype(im.dtype)
    return out.reshape(size)
