"""CPU: the product's seed tree (csrc/kdtree.cpp build + ia3_kdtree.h query, compiled for the host by
tests/native/kd_cpu.cpp) against scipy.spatial.cKDTree — the library the reference asks for its Voronoi cells
(External/Fitting_v4.py:601,612, :422-424).  Exact Voronoi ties go to the seed the query meets first, so the point
permutation, the node layout and the traversal order must all be scipy's; integer seed coordinates (what get_seeds
returns) make ties and equal split coordinates the normal case."""
import os
import subprocess
import ctypes as C
import numpy as np
import pytest
from scipy.spatial import cKDTree

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "native", "libkdcpu.so")


@pytest.fixture(scope="module")
def kd():
    src = os.path.join(HERE, "native", "kd_cpu.cpp")
    dep = os.path.join(HERE, "..", "imageanalysis3_amd", "csrc")
    newest = max(os.path.getmtime(p) for p in (src, os.path.join(dep, "kdtree.cpp"), os.path.join(dep, "ia3_kdtree.h")))
    if not os.path.isfile(SO) or os.path.getmtime(SO) < newest:
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", SO, src])
    return C.CDLL(SO)


def fields(rng, trial):
    n = int(rng.integers(2, 6000))
    kind = trial % 4
    if kind == 0:      # isolated-like integer seeds in a production-size stack
        pts = np.floor(rng.uniform(0, [50, 2048, 2048], (n, 3)))
    elif kind == 1:    # territories (SURVEY.md §8d layout B)
        terr = rng.uniform([10, 100, 100], [40, 1900, 1900], (200, 3))
        pts = np.floor(terr[rng.integers(0, 200, n)] + rng.normal(0, [5, 15, 15], (n, 3)))
    elif kind == 2:    # one dense blob: many equal coordinates, many exact ties
        pts = np.floor(rng.normal([25, 300, 300], [3, 6, 6], (n, 3)))
    else:              # arbitrary float centres (the class API takes them)
        pts = rng.uniform(0, 100, (n, 3))
    if kind != 3:
        pts = np.unique(pts, axis=0)
        rng.shuffle(pts)
    return np.ascontiguousarray(pts, dtype=np.float64)


def flatten(node, out):
    out.append((node.split_dim, node.split if node.split_dim >= 0 else 0.0, node.start_idx, node.end_idx))
    if node.split_dim >= 0:
        flatten(node.lesser, out)
        flatten(node.greater, out)


def build(kd, pts):
    n = len(pts)
    idx = np.zeros(n, np.int32)
    nodes = np.zeros((2 * n + 8, 7))
    nn = kd.ia3cpu_kd_build(pts.ctypes.data_as(C.c_void_p), n, idx.ctypes.data_as(C.c_void_p),
                            nodes.ctypes.data_as(C.c_void_p), len(nodes))
    return idx, nodes[:nn]


def query(kd, q, upper, stride=1, cap=64):
    q = np.ascontiguousarray(q, dtype=np.float64)
    oi = np.zeros(len(q), np.int32)
    od = np.zeros(len(q))
    ovf = kd.ia3cpu_kd_query(q.ctypes.data_as(C.c_void_p), len(q), C.c_double(upper), stride, cap,
                             oi.ctypes.data_as(C.c_void_p), od.ctypes.data_as(C.c_void_p), None)
    return oi, od, ovf


def test_build_layout_equals_scipy(kd):
    rng = np.random.default_rng(11)
    for trial in range(48):
        pts = fields(rng, trial)
        t = cKDTree(pts)
        idx, nodes = build(kd, pts)
        assert np.array_equal(idx, t.indices), trial
        ref = []
        flatten(t.tree, ref)
        assert len(ref) == len(nodes), trial
        for r, m in zip(ref, nodes):
            assert r[0] == int(m[0]) and r[2] == int(m[2]) and r[3] == int(m[3]), trial
            if r[0] >= 0:
                assert r[1] == m[1], trial
        # parents consistent with children
        for k, m in enumerate(nodes):
            if int(m[0]) >= 0:
                assert int(nodes[int(m[4])][6]) == k and int(nodes[int(m[5])][6]) == k


def test_query_equals_scipy_on_ball_voxels(kd):
    """Every voxel of the 512-voxel ball around a few hundred seeds: the winner of tree.query, ties included."""
    rng = np.random.default_rng(12)
    off = np.stack(np.meshgrid(*[np.arange(-5, 5)] * 3, indexing="ij"), -1).reshape(-1, 3)
    off = off[(off ** 2).sum(1) <= 25]
    assert len(off) == 512
    n_ties = 0
    for trial in range(24):
        pts = fields(rng, trial)
        t = cKDTree(pts)
        build(kd, pts)
        sel = rng.integers(0, len(pts), 300)
        q = (np.trunc(pts[sel])[:, None, :] + off[None]).reshape(-1, 3)
        d, i = t.query(q, distance_upper_bound=10.0)
        oi, od, ovf = query(kd, q, 10.0, stride=64, cap=32)
        assert not ovf
        assert np.array_equal(oi, i), trial
        found = i < len(pts)
        assert np.array_equal(od[found], (d[found] ** 2)) or np.allclose(od[found], d[found] ** 2, rtol=1e-15, atol=0)
        if len(pts) <= 1500:
            dd = ((q[:, None, :] - pts[None, :, :]) ** 2).sum(-1)
            n_ties += int(((dd == dd.min(1)[:, None]).sum(1) > 1).sum())
    assert n_ties > 1000   # the comparison above is about ties


def test_queue_overflow_is_reported(kd):
    rng = np.random.default_rng(13)
    pts = np.unique(np.floor(rng.normal([25, 300, 300], [3, 6, 6], (5000, 3))), axis=0)
    pts = np.ascontiguousarray(pts)
    build(kd, pts)
    q = pts[:2000] + 0.0
    oi, od, ovf = query(kd, q, 10.0, stride=1, cap=1)
    ok = oi >= 0
    i = cKDTree(pts).query(q, distance_upper_bound=10.0)[1]
    assert np.array_equal(oi[ok], i[ok])   # a query is either exact or flagged
