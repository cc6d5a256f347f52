"""CPU: the product's LM core (csrc/ia3_lm.h + ia3_model.h + ia3_init.h, compiled for the host by
tests/native/lm_cpu.cpp) against MINPACK (scipy.optimize.leastsq through the oracle) and the golden rows.

This checks the ALGORITHM the HIP kernel runs (lmder control flow on the normal equations, model,
Jacobian, start point) without a GPU; the -m gpu tests check the kernel itself."""
import os
import subprocess
import ctypes as C
import numpy as np
import pytest
from conftest import build_case, load_golden
import np_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "native", "liblmcpu.so")


@pytest.fixture(scope="module")
def lm():
    src = os.path.join(HERE, "native", "lm_cpu.cpp")
    if not os.path.isfile(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", SO, src])
    return C.CDLL(SO)


def cfit(lm, vals, X, center, delta, kind):
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    coords = np.ascontiguousarray(np.array(X).T, dtype=np.int32)
    p = np.zeros(11, np.float32)
    x = np.zeros(10)
    info = np.zeros(3, np.int32)
    c = np.array(center, dtype=np.float64)
    rc = lm.ia3cpu_gaussfit(vals.ctypes.data_as(C.c_void_p), coords.ctypes.data_as(C.c_void_p), C.c_int(len(vals)),
                            c.ctypes.data_as(C.c_void_p), C.c_double(delta), C.c_double(0.5), C.c_double(4.0),
                            C.c_double(1.5), C.c_int(kind), p.ctypes.data_as(C.c_void_p),
                            x.ctypes.data_as(C.c_void_p), info.ctypes.data_as(C.c_void_p))
    return rc, p, x, info


@pytest.mark.parametrize("name", ["c1_f32", "c1_u16", "edge_f32", "clu_f32"])
def test_first_fit_rows_match_minpack(lm, name):
    g = load_golden("fit_%s.npz" % name)
    im = build_case(name)
    seeds = O.get_seeds(im, th_seed=600)
    f = O.iter_fit_seed_points(im, seeds.T)
    f.firstfit()
    kind = 1 if im.dtype == np.uint16 else 0
    worst = 0.0
    for ic, (im_, X, center) in enumerate(f.gparms):
        rc, p, x, info = cfit(lm, im_, X, center, 1.0, kind)
        ref = np.asarray(f.ps[ic], dtype=np.float64)
        if rc:
            assert np.isnan(ref).all()
            continue
        rel = np.abs(p[:8] - ref[:8]) / np.abs(ref[:8])
        worst = max(worst, rel.max())
        assert info[0] in (1, 2, 3), info
    assert worst <= 1e-6, worst


def test_nfev_matches_minpack(lm):
    """Same iteration path as MINPACK: the number of function evaluations agrees fit by fit."""
    from scipy.optimize import leastsq
    im = build_case("c1_f32")
    seeds = O.get_seeds(im, th_seed=600)
    f = O.iter_fit_seed_points(im, seeds.T)
    f.firstfit()
    same = 0
    for ic, (im_, X, center) in enumerate(f.gparms[:25]):
        obj = O.GaussianFit(im_, X, center=center, delta_center=1.0)
        _, _, infodict, _, _ = leastsq(obj.calc_eps, obj.p_, Dfun=obj.calc_jac, maxfev=1000, full_output=True)
        rc, p, x, info = cfit(lm, im_, X, center, 1.0, 0)
        same += int(info[1] == infodict["nfev"])
    assert same >= 23, same


def test_too_few_voxels_fails(lm):
    rc, p, x, info = cfit(lm, np.arange(9.0), np.zeros((3, 9), int), [0, 0, 0], 1.0, 0)
    assert rc == 1


def test_legacy_v3_model_matches_reference_rows(lm):
    """FitCfg::variant 1 (Fitting_v3's to_center and per-axis start widths) against the reference's own
    first-fit rows of the legacy golden case."""
    from conftest import build_legacy
    im, m = build_legacy()
    g = load_golden("legacy.npz")
    sa = tuple(m["seeding"]["default"][:-1]) + (False,)
    iw = np.array([np.log((16.0 - w * w) / (w * w - 0.25)) for w in (1.35, 1.9, 1.9)])
    for i in (0, 3):
        seeds = O.legacy_get_seed_in_distance(im, g["coords"][i], *sa)
        f = O.iter_fit_seed_points_v3(im, seeds.T, *m["fitting_args"])
        zb, xb, yb = O.ball_offsets(5)
        for ic, (zc, xc, yc) in enumerate(f.centers):
            z, x, y = O._in_dim(int(zc) + zb, int(xc) + xb, int(yc) + yb, *im.shape)
            X_full = np.array([z, x, y], dtype=int)
            X = X_full[:, f._nearest_is_me(X_full, ic)]
            vals = im[X[0], X[1], X[2]]
            p = np.zeros(11, np.float32); xo = np.zeros(10); info = np.zeros(3, np.int32)
            v = np.ascontiguousarray(vals, dtype=np.float64)
            co = np.ascontiguousarray(X.T, dtype=np.int32)
            c = np.array([zc, xc, yc], dtype=np.float64)
            rc = lm.ia3cpu_gaussfit_v3(v.ctypes.data_as(C.c_void_p), co.ctypes.data_as(C.c_void_p), C.c_int(len(v)),
                                       c.ctypes.data_as(C.c_void_p), C.c_double(1.0), iw.ctypes.data_as(C.c_void_p),
                                       C.c_int(1), p.ctypes.data_as(C.c_void_p), xo.ctypes.data_as(C.c_void_p),
                                       info.ctypes.data_as(C.c_void_p))
            assert rc == 0
            ref = g["first_%d" % i][ic].astype(np.float64)
            rel = np.abs(p[:8] - ref[:8]) / np.abs(ref[:8])
            assert rel.max() <= 1e-5, (i, ic, rel, info)
