"""GPU parity tests: the HIP path (through the C ABI / drop-in shims) against the committed golden
fixtures produced by the reference, and against the NumPy oracle on the same seeded inputs.

Tolerances (BASELINE.json north_star): integer / index work bit-exact (filters in the stack dtype,
seed coordinates); fitted (h, z, x, y, bk, sigma_z, sigma_x, sigma_y) within 1e-4 relative.
"""
import os
import zlib
import numpy as np
import pytest
from conftest import build_case, load_golden

pytestmark = pytest.mark.gpu

FIT_CASES = ["c1_f32", "c1_u16", "m_f32", "edge_f32", "hot_u16"]
# crowded layouts: exact Voronoi ties, resolved as the reference's cKDTree does (first-fit voxel counts are exact)
CROWDED_CASES = ["clu_f32", "club_f32"]
RTOL = 1e-4


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def seed_set(s):
    """canonical form of an (N,4) seed table: rows sorted by coordinate (ties in h may be permuted)."""
    s = np.asarray(s)
    if len(s) == 0:
        return s.reshape(0, 4)
    return s[np.lexsort((s[:, 2], s[:, 1], s[:, 0]))]


def match_rows(a, b, tol=0.05):
    """pair rows of two spot tables by fitted centre; returns index arrays (ia, ib)."""
    from scipy.spatial import cKDTree
    assert len(a) == len(b), (len(a), len(b))
    if len(a) == 0:
        return np.zeros(0, int), np.zeros(0, int)
    d, j = cKDTree(b[:, 1:4]).query(a[:, 1:4])
    assert (d < tol).all(), d.max()
    assert len(np.unique(j)) == len(j)
    return np.arange(len(a)), j


def assert_rows_close(a, b, rtol=RTOL):
    ia, ib = match_rows(a, b)
    a, b = a[ia].astype(np.float64), b[ib].astype(np.float64)
    rel = np.abs(a[:, :8] - b[:, :8]) / np.abs(b[:, :8])
    assert rel.max() <= rtol, ("max rel err %g at %s" % (rel.max(), np.unravel_index(rel.argmax(), rel.shape)))
    # sines of the rotation angles are ~0 for axis-aligned spots: absolute tolerance
    assert np.abs(a[:, 8:10] - b[:, 8:10]).max() <= 2e-3
    assert (np.abs(a[:, 10] - b[:, 10]) / np.abs(b[:, 10])).max() <= 1e-3


# ---------------------------------------------------------------------------------------------
# filters
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["c1_f32", "c1_u16", "edge_f32"])
@pytest.mark.parametrize("sigma,truncate,mode", [(0.75, 4.0, "reflect"), (7.5, 4.0, "reflect"), (3, 2, "nearest"),
                                                 (5, 2, "nearest"), (2.0, 4.0, "reflect"), (1.3, 3.0, "nearest")])
def test_gaussian_filter_bit_exact(name, sigma, truncate, mode):
    import np_oracle as O
    from imageanalysis3_amd.correction_tools.filter import gaussian_filter
    im = build_case(name)
    got = gaussian_filter(im, sigma, mode=mode, truncate=truncate)
    ref = O.gaussian_filter(im, sigma, mode=mode, truncate=truncate)
    assert got.dtype == im.dtype and got.shape == im.shape
    assert np.array_equal(got, ref), "%d voxels differ" % (got != ref).sum()


@pytest.mark.parametrize("shape", [(7, 45, 83), (33, 70, 130), (5, 31, 65), (64, 33, 257), (3, 8, 8)])
@pytest.mark.parametrize("dtype", [np.float32, np.uint16])
def test_gaussian_filter_ragged_shapes_bit_exact(shape, dtype):
    """Partial tiles of the fused short-filter kernel, partial segments / transposing tiles of the long one, axes
    shorter than the filter radius (the border map wraps more than once), every fixed radius and the generic path."""
    from scipy import ndimage as ndi
    from imageanalysis3_amd.correction_tools.filter import gaussian_filter
    rng = np.random.RandomState(shape[1] * 7 + shape[2])
    a = rng.gamma(2.0, 300.0, size=shape)
    im = a.astype(np.float32) if dtype == np.float32 else np.clip(a, 0, 65535).astype(np.uint16)
    for sigma, truncate, mode in [(0.75, 4.0, "reflect"), (0.75, 4.0, "nearest"), (7.5, 4.0, "reflect"),
                                  (7.5, 4.0, "nearest"), (3, 2, "nearest"), (5, 2, "reflect"), (1.1, 4.0, "reflect")]:
        got = gaussian_filter(im, sigma, mode=mode, truncate=truncate)
        ref = ndi.gaussian_filter(im, sigma, mode=mode, truncate=truncate)
        assert got.dtype == im.dtype
        assert np.array_equal(got, ref), (shape, sigma, mode, int((got != ref).sum()))


def test_gaussian_filter_matches_scipy_directly():
    from scipy import ndimage as ndi
    from imageanalysis3_amd.correction_tools.filter import gaussian_filter
    im = build_case("c1_f32")
    assert np.array_equal(gaussian_filter(im, 0.75), ndi.gaussian_filter(im, 0.75))
    assert np.array_equal(gaussian_filter(im, 7.5), ndi.gaussian_filter(im, 7.5))


@pytest.mark.parametrize("name", ["c1_f32", "c1_u16", "hot_u16"])
def test_highpass_and_hot_pixels_golden(name):
    from imageanalysis3_amd.correction_tools.filter import gaussian_high_pass_filter, Remove_Hot_Pixels
    g = load_golden("filters.npz")
    im = build_case(name)
    keep = im.copy()
    for sg in (3, 5):
        hp = gaussian_high_pass_filter(im, sg, 2)
        assert hp.dtype == im.dtype
        assert crc(hp) == g["hp_%s_s%d_crc" % (name, sg)]
    rh = Remove_Hot_Pixels(im, dtype=im.dtype)
    assert crc(rh) == g["rhp_%s_crc" % name]
    assert np.array_equal(im, keep)  # inputs are never mutated


# ---------------------------------------------------------------------------------------------
# seeding
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", FIT_CASES + CROWDED_CASES)
def test_get_seeds_golden(name):
    from imageanalysis3_amd.spot_tools.fitting import get_seeds
    g = load_golden("fit_%s.npz" % name)
    im = build_case(name)

    def same(got, key):
        ref = g[key]
        assert got.dtype == np.float64 and got.shape == ref.shape, (key, got.shape, ref.shape)
        assert np.array_equal(seed_set(got), seed_set(ref)), key
        assert (np.diff(got[:, 3]) <= 0).all(), key  # brightest first

    same(get_seeds(im, th_seed=600, return_h=True), "seeds_h")
    same(get_seeds(im, th_seed=600, use_dynamic_th=False, return_h=True), "seeds_nodyn")
    same(get_seeds(im, th_seed=9000, return_h=True, min_dynamic_seeds=5), "seeds_hi_th")
    same(get_seeds(im, th_seed=600, remove_hot_pixel=False, return_h=True), "seeds_nohot")
    same(get_seeds(im, th_seed=600, sel_center=list(g["sel_center"]), seed_radius=20, return_h=True), "seeds_sel")
    same(get_seeds(im, th_seed=600, min_edge_distance=0, return_h=True), "seeds_edge0")
    top = get_seeds(im, th_seed=600, max_num_seeds=10, return_h=True)
    ref = g["seeds_top10"]
    assert top.shape == ref.shape
    assert np.array_equal(np.sort(top[:, 3]), np.sort(ref[:, 3]))
    s3 = get_seeds(im, th_seed=600)
    assert s3.shape == (len(g["seeds_h"]), 3)


@pytest.mark.parametrize("shape,dtype", [((9, 45, 83), np.float32), ((30, 70, 130), np.uint16), ((12, 33, 257), np.float32)])
def test_get_seeds_ragged_shapes_vs_oracle(shape, dtype):
    """Tile edges of the detector and of the fused filter on shapes that are no multiple of anything."""
    import np_oracle as O
    from imageanalysis3_amd import synth
    from imageanalysis3_amd.spot_tools.fitting import get_seeds
    im, c, h = synth.make_fov(shape, 12, 77, dtype=dtype, margin=(2, 5, 5), min_sep=6.0)
    for kw in (dict(th_seed=600), dict(th_seed=300, min_edge_distance=0, remove_hot_pixel=False),
               dict(th_seed=5000, use_dynamic_th=True, min_dynamic_seeds=8), dict(th_seed=600, max_num_seeds=5)):
        got = get_seeds(im, return_h=True, **kw)
        ref = O.get_seeds(im, return_h=True, **kw)
        assert got.shape == ref.shape, (kw, got.shape, ref.shape)
        if "max_num_seeds" in kw:
            assert np.array_equal(np.sort(got[:, 3]), np.sort(ref[:, 3]))
        else:
            assert np.array_equal(seed_set(got), seed_set(ref)), kw


def test_get_seeds_many_candidates_host_finish_path():
    """More than 8192 candidates: the device finish declines and the host tail takes over; same seeds either way."""
    import np_oracle as O
    from imageanalysis3_amd.spot_tools.fitting import get_seeds
    from imageanalysis3_amd import synth
    im = synth.make_fov((40, 160, 200), 20, 5)[0]
    for kw in (dict(th_seed=-1000.0, use_dynamic_th=False, remove_hot_pixel=False, min_edge_distance=0),
               dict(th_seed=-1000.0, use_dynamic_th=False, remove_hot_pixel=True, hot_pixel_th=3),
               dict(th_seed=-500.0, use_dynamic_th=True, dynamic_niters=4, min_dynamic_seeds=100000)):
        got = get_seeds(im, return_h=True, **kw)
        ref = O.get_seeds(im, return_h=True, **kw)
        assert len(ref) > 8192 or kw.get("remove_hot_pixel"), len(ref)
        assert got.shape == ref.shape, (kw, got.shape, ref.shape)
        assert np.array_equal(seed_set(got), seed_set(ref)), kw


def test_get_seeds_errors_and_empty():
    from imageanalysis3_amd.spot_tools.fitting import get_seeds
    with pytest.raises(TypeError):
        get_seeds([[1, 2], [3, 4]])
    im = build_case("c1_f32")
    with pytest.raises(IndexError):
        get_seeds(im, sel_center=[1, 2])
    flat = np.full((12, 32, 32), 400, dtype=np.uint16)
    s = get_seeds(flat, th_seed=600)
    assert s.shape == (0, 3)


# ---------------------------------------------------------------------------------------------
# fitting
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", FIT_CASES + CROWDED_CASES)
def test_first_and_final_fit_golden(name):
    from imageanalysis3_amd.External.Fitting_v4 import iter_fit_seed_points
    g = load_golden("fit_%s.npz" % name)
    im = build_case(name)
    seeds = g["seeds_h"][:, :3]  # the reference's own seed order
    f = iter_fit_seed_points(im, seeds.T, radius_fit=5)
    f.firstfit()
    first = np.array(f.ps)
    assert first.dtype == np.float32 and first.shape == g["first_ps"].shape
    assert np.array_equal(f.nvox, g["first_nvox"])
    assert_rows_close(first, g["first_ps"])
    f.repeatfit()
    assert f.n_iter == int(g["n_iter"])
    assert_rows_close(np.array(f.ps), g["final_ps"])


@pytest.mark.parametrize("name", FIT_CASES)
def test_fit_fov_image_golden(name):
    from imageanalysis3_amd.spot_tools.fitting import fit_fov_image
    g = load_golden("fit_%s.npz" % name)
    im = build_case(name)
    keep = im.copy()
    t = fit_fov_image(im, "647", th_seed=600, max_num_seeds=None, verbose=False)
    assert t.dtype == np.float32 and t.shape == g["table"].shape
    assert_rows_close(t, g["table"])
    assert np.array_equal(im, keep)
    t20 = fit_fov_image(im, "647", th_seed=600, max_num_seeds=20, verbose=False)
    assert t20.shape == g["table_max20"].shape
    if im.dtype == np.float32:   # no equal DoG heights in the float32 cases: the 20 brightest seeds are the same 20
        assert_rows_close(t20, g["table_max20"])
    else:                        # uint16 heights tie (NumPy's introsort order among equals is implementation-defined):
        #                          every row must still be one of the full table's spots
        from scipy.spatial import cKDTree
        d, j = cKDTree(g["table"][:, 1:4]).query(t20[:, 1:4])
        assert d.max() < 0.05, d.max()


@pytest.mark.parametrize("name", ["c1_f32", "c1_u16", "clu_f32"])
def test_percentile_threshold_and_seed_mask_golden(name):
    """Options that run host-side in the shim — the percentile threshold of get_seeds (spot_tools/fitting.py:75-76) and
    fit_fov_image's seed_mask (:210-218) — against the reference's own outputs (tests/golden/seedopts.npz)."""
    from conftest import seed_mask_for
    from imageanalysis3_amd.spot_tools.fitting import fit_fov_image, get_seeds
    from imageanalysis3_amd import _lib as L
    g = load_golden("seedopts.npz")
    im = build_case(name)
    for per in (95, 99.5, 98):
        tag = "%s_per%s" % (name, str(per).replace(".", "p"))
        got = get_seeds(im, use_percentile=True, th_seed_per=per, return_h=True)
        assert np.array_equal(seed_set(got), seed_set(g[tag])), tag
        got = get_seeds(im, use_percentile=True, th_seed_per=per, use_dynamic_th=False, return_h=True)
        assert np.array_equal(seed_set(got), seed_set(g[tag + "_nodyn"])), tag
    got = get_seeds(im, use_percentile=True, th_seed_per=99.5, return_h=True, sel_center=[s // 2 for s in im.shape],
                    seed_radius=25)
    assert np.array_equal(seed_set(got), seed_set(g[name + "_per_sel"]))
    t = fit_fov_image(im, "647", use_percentile=True, th_seed_per=99.5, max_num_seeds=None, verbose=False)
    assert t.shape == g[name + "_per_table"].shape
    assert_rows_close(t, g[name + "_per_table"])
    with L.DeviceStack.upload(im) as st:   # the same through a resident stack (the percentile needs the host copy)
        t = fit_fov_image(st, "647", use_percentile=True, th_seed_per=99.5, max_num_seeds=None, verbose=False)
    assert_rows_close(t, g[name + "_per_table"])
    mask = seed_mask_for(im.shape)
    t = fit_fov_image(im, "647", th_seed=600, max_num_seeds=None, seed_mask=mask, verbose=False)
    assert t.shape == g[name + "_mask_table"].shape
    assert_rows_close(t, g[name + "_mask_table"])
    seeds = load_golden("fit_%s.npz" % name)["seeds_h"]
    t = fit_fov_image(im, "647", seeds=seeds, seed_mask=mask > 0, verbose=False)
    assert t.shape == g[name + "_mask_given"].shape
    assert_rows_close(t, g[name + "_mask_given"])


def test_single_spot_known_answer():
    from imageanalysis3_amd import synth
    from imageanalysis3_amd.External.Fitting_v4 import iter_fit_seed_points
    g = load_golden("single_spot.npz")
    im64 = np.full(tuple(g["shape"]), 100.0)
    synth.add_spots(im64, g["center"], np.array([2000.0]))
    f = iter_fit_seed_points(im64.astype(np.float32), np.array([[14.0], [31.0], [33.0]]), radius_fit=5)
    f.firstfit()
    f.repeatfit()
    ps = np.array(f.ps)
    assert_rows_close(ps, g["ps"])
    assert np.allclose(ps[0, 1:4], g["center"][0], atol=2e-3)
    assert f.n_iter == int(g["n_iter"])


def test_clustered_field_vs_oracle_and_reference():
    """Crowded layout.  The reference's cKDTree gives a voxel that is exactly as far from two seeds to the one its query
    meets first; the library resolves those ties with the same tree layout and traversal (csrc/kdtree.cpp,
    ia3_kdtree.h, voronoi_ties_k), so the tables must equal the reference's own goldens — the 30 x 128 x 128 crowded
    case and a 50 x 256 x 256 layout-B field — to the tolerance of every other fit, and the oracle (scipy's cKDTree) to
    float32 rounding: this also exercises the ordered Gauss-Seidel sweeps across overlapping balls."""
    import np_oracle as O
    from imageanalysis3_amd.spot_tools.fitting import fit_fov_image
    for name in ("clu_f32", "club_f32"):
        g = load_golden("fit_%s.npz" % name)
        im = build_case(name)
        t = fit_fov_image(im, "647", th_seed=600, max_num_seeds=None, verbose=False)
        o = O.fit_fov_image(im, "647", th_seed=600, max_num_seeds=None)
        assert t.shape == o.shape == g["table"].shape, name
        assert_rows_close(t, o, rtol=1e-6)
        relg = np.abs(t[:, :8].astype(float) - g["table"][:, :8]) / np.abs(g["table"][:, :8])
        assert relg.max() <= 1e-4, (name, relg.max())


def test_voronoi_tie_query_overflow_is_finished_on_the_host():
    """A device tie query that runs out of queue entries (24 per voxel; never seen on real fields) no longer fails the
    call: the host repeats the tie queries with the same tree and an unbounded queue.  IA3_TUNE_FIT_KDQ = 1 makes
    nearly every query of the crowded goldens overflow: tables identical to the default run, bit for bit."""
    import ctypes as C
    from imageanalysis3_amd import _lib as L
    from imageanalysis3_amd.spot_tools.fitting import fit_fov_image
    from imageanalysis3_amd.External.Fitting_v4 import iter_fit_seed_points
    for name in ("clu_f32", "club_f32"):
        g = load_golden("fit_%s.npz" % name)
        im = build_case(name)
        ref = fit_fov_image(im, "647", th_seed=600, max_num_seeds=None, verbose=False)
        try:
            L.check(L.lib().ia3_set_tuning(C.c_int(13), C.c_int(1)))
            got = fit_fov_image(im, "647", th_seed=600, max_num_seeds=None, verbose=False)
            f = iter_fit_seed_points(im, g["seeds_h"][:, :3].T, radius_fit=5)
            f.firstfit()
            assert np.array_equal(f.nvox, g["first_nvox"]), name       # the Voronoi cells are the reference's
        finally:
            L.check(L.lib().ia3_set_tuning(C.c_int(13), C.c_int(24)))
        assert np.array_equal(got, ref), name
        assert got.shape == g["table"].shape


def test_dependency_wait_abort_is_reported_and_the_library_recovers():
    """The refit sweeps wait for the fits they depend on with a bound (2^22 polls, ~28 s) behind which the launch gives up
    instead of hanging the device.  IA3_DEBUG_FIT_WAITBOUND lowers it to a few polls, so that an ordinary wait in a
    crowded field trips it: the call fails with the library's error (no table), and the next call — bound restored —
    returns the golden table."""
    import ctypes as C
    from imageanalysis3_amd import _lib as L
    from imageanalysis3_amd.spot_tools.fitting import fit_fov_image
    g = load_golden("fit_club_f32.npz")
    im = build_case("club_f32")
    try:
        L.check(L.lib().ia3_set_tuning(C.c_int(101), C.c_int(16)))
        with pytest.raises(L.IA3Error, match="dependency wait exceeded its bound"):
            fit_fov_image(im, "647", th_seed=600, max_num_seeds=None, verbose=False)
    finally:
        L.check(L.lib().ia3_set_tuning(C.c_int(101), C.c_int(0)))
    t = fit_fov_image(im, "647", th_seed=600, max_num_seeds=None, verbose=False)
    assert t.shape == g["table"].shape
    assert_rows_close(t, g["table"])


def test_refit_with_unchanged_neighbours_is_not_repeated():
    """A repeat fit is a function of the image ball and of the overlapping seeds' records; when those have not changed
    since the seed's previous refit, the sweep would reproduce that fit bit for bit (centre unchanged -> converged), so it
    is not run (IA3_TUNE_FIT_MEMO).  Same tables, same sweep counts with the shortcut on and off — isolated, crowded and
    uint16 fields (plateau twins refitting noise to maxfev are what it is for) — and fewer evaluations with it."""
    import ctypes as C
    from imageanalysis3_amd import synth, _lib as L
    shape = (30, 160, 160)
    ims = [synth.make_fov(shape, 60 if k % 2 == 0 else 90, 20 + k, layout="isolated" if k % 2 == 0 else "clustered", n_territories=5)[0]
           for k in range(4)]
    ims += [im.astype(np.uint16) for im in ims] + [build_case("club_f32"), build_case("hot_u16")]
    sp, keep = L.make_seed_params(600.0, max_num_seeds=None)
    fp = L.make_fit_params()
    res = {}
    try:
        for memo in (1, 0):
            L.check(L.lib().ia3_set_tuning(C.c_int(14), C.c_int(memo)))
            res[memo] = [L.fit_fovs([im], sp, fp, in_flight=1) for im in ims]
            res[memo].append(L.fit_fovs(ims[4:8], sp, fp, in_flight=4))     # the uint16 ones as one group fit
    finally:
        L.check(L.lib().ia3_set_tuning(C.c_int(14), C.c_int(1)))
    saved = 0
    for (t1, i1), (t0, i0) in zip(res[1], res[0]):
        for a, b, ia, ib in zip(t1, t0, i1, i0):
            assert np.array_equal(a, b)
            assert ia["n_iter"] == ib["n_iter"] and ia["n_seeds"] == ib["n_seeds"]
            assert ia["nfev"] <= ib["nfev"] and ia["fits"] <= ib["fits"]
            saved += ib["fits"] - ia["fits"]
    assert saved > 0


def test_gauss_seidel_order_mid_size_exact():
    """333 seeds in 16 territories (218 overlapping pairs, 6 sweeps): the dependency-ordered kernel must reproduce
    the sequential reference order exactly — same n_iter, float32-identical rows — and do so on every run."""
    import np_oracle as O
    from imageanalysis3_amd import synth
    from imageanalysis3_amd.External.Fitting_v4 import iter_fit_seed_points
    from imageanalysis3_amd.spot_tools.fitting import get_seeds
    im, c, h = synth.make_fov((50, 512, 512), 400, 3, layout="clustered", n_territories=16)
    seeds = get_seeds(im, th_seed=600.0)
    assert np.array_equal(seeds, O.get_seeds(im, th_seed=600.0))
    fo = O.iter_fit_seed_points(im, seeds.T)   # Voronoi ties as the reference's cKDTree leaves them
    fo.firstfit()
    fo.repeatfit()
    po = np.array(fo.ps, dtype=np.float64)
    assert fo.n_iter >= 4   # the case really needs several sweeps
    for rep in range(2):
        f = iter_fit_seed_points(im, seeds.T)
        f.firstfit()
        f.repeatfit()
        p = np.array(f.ps, dtype=np.float64)
        assert f.n_iter == fo.n_iter
        rel = np.abs(p[:, :8] - po[:, :8]) / np.abs(po[:, :8])
        assert np.nanmax(rel) <= 1e-6, np.nanmax(rel)


def test_fit_edge_cases():
    from imageanalysis3_amd.spot_tools.fitting import fit_fov_image
    from imageanalysis3_amd.External.Fitting_v4 import iter_fit_seed_points
    flat = np.full((12, 32, 32), 400, dtype=np.uint16)
    assert fit_fov_image(flat, "647", th_seed=600, verbose=False).size == 0
    # no seeds given to the fitter: loops are no-ops, n_iter = 0
    f = iter_fit_seed_points(flat, np.zeros((3, 0)))
    f.firstfit()
    f.repeatfit()
    assert f.n_iter == 0 and len(f.ps) == 0
    # a seed in the corner of a tiny image: ball clipped, still >= 10 voxels
    im = build_case("edge_f32")
    f = iter_fit_seed_points(im, np.array([[0.0], [0.0], [0.0]]))
    f.firstfit()
    assert f.nvox[0] == sum(1 for z in range(5) for x in range(5) for y in range(5) if z * z + x * x + y * y <= 25)


def test_centers_and_sparse_golden():
    from imageanalysis3_amd.spot_tools.fitting import get_centers, select_sparse_centers
    g = load_golden("fit_c1_f32.npz")
    im = build_case("c1_f32")
    c = get_centers(im, th_seed=600)
    assert c.shape == g["centers"].shape
    from scipy.spatial import cKDTree
    d, j = cKDTree(g["centers"]).query(c)
    assert d.max() < 1e-3
    sp = select_sparse_centers(g["centers"], distance_th=25)
    assert np.array_equal(sp, g["sparse"])


# ---------------------------------------------------------------------------------------------
# drift
# ---------------------------------------------------------------------------------------------
def _bead_pair():
    from imageanalysis3_amd import synth
    g = load_golden("drift.npz")
    ref, src, bc, bh = synth.make_bead_pair(tuple(g["bead_shape"]), 120, 21, g["bead_true_d"])
    return g, ref, src


def test_fft3d_from2d_golden():
    from imageanalysis3_amd.alignment_tools import fft3d_from2d, fftalign_2d
    g, ref, src = _bead_pair()
    assert np.array_equal(fft3d_from2d(src, ref, gb=0, max_disp=128), g["fft3d"])
    assert np.array_equal(np.array(fftalign_2d(np.max(src, 0), np.max(ref, 0), max_disp=50)), g["fft3d_F4style_xy"])
    # u16 twin and a window that excludes the true peak
    import np_oracle as O
    s16, r16 = src.astype(np.uint16), ref.astype(np.uint16)
    assert np.array_equal(fft3d_from2d(s16, r16, gb=0, max_disp=128), O.fft3d_from2d(s16, r16, gb=0, max_disp=128))
    assert np.array_equal(np.array(fftalign_2d(np.max(src, 0), np.max(ref, 0), max_disp=3)),
                          np.array(O.fftalign_2d(np.max(src, 0), np.max(ref, 0), max_disp=3)))


def test_align_image_bead_path_golden():
    from imageanalysis3_amd.correction_tools.alignment import align_image, generate_drift_crops
    g, ref, src = _bead_pair()
    for k in range(4):
        assert np.array_equal(generate_drift_crops(list(g["crops_%d_size" % k])), g["crops_%d" % k])
    drift, flag = align_image(src, ref, use_autocorr=False, verbose=False)
    assert flag == int(g["align_beads_flag"])
    assert np.allclose(drift, g["align_beads_drift"], atol=2e-4)
    assert np.allclose(drift, -g["bead_true_d"], atol=0.02)


def test_align_image_resident_inputs_equal_host_inputs():
    from imageanalysis3_amd import synth, _lib as L
    from imageanalysis3_amd.correction_tools.alignment import align_image
    shape = (24, 256, 256)
    ref, src, c, h = synth.make_bead_pair(shape, 60, 9, (0.4, -2.3, 3.1), margin=(4, 12, 12), min_sep=10.0)
    a, b = L.DeviceStack.upload(src), L.DeviceStack.upload(ref)
    try:
        for auto in (True, False):
            d0, f0 = align_image(src, ref, use_autocorr=auto, verbose=False, correction_args={'single_im_size': shape})
            d1, f1 = align_image(a, b, use_autocorr=auto, verbose=False, correction_args={'single_im_size': shape})
            assert f0 == f1 and np.array_equal(d0, d1)
    finally:
        a.free(); b.free()


def test_drift_reference_and_batched_crops_equal_the_per_crop_function():
    """align_image's phase-correlation path queues the whole chain of a crop without a host round trip (coarse peak ->
    DFT offsets on the device), the first three crops back to back, and with a DriftReference the reference crops'
    spectra are made once: all of it must give the per-crop function's shifts (phase_cross_correlation on the crops, two
    waits per crop) bit for bit, for uint16 and float32 stacks, both normalisations, and when the rule needs more than
    three crops."""
    from imageanalysis3_amd import synth, _lib as L
    from imageanalysis3_amd.correction_tools import alignment as A
    shape = (24, 256, 256)
    ref, src, c, h = synth.make_bead_pair(shape, 60, 9, (0.4, -2.3, 3.1), margin=(4, 12, 12), min_sep=10.0)
    crops = A.generate_drift_crops(shape)
    for dtype in (np.uint16, np.float32):
        r, s_ = ref.astype(dtype), src.astype(dtype)
        with L.DeviceStack.upload(r) as dr, L.DeviceStack.upload(s_) as ds:
            for norm in (None, "phase"):
                old = A.DEFAULT_NORMALIZATION
                A.DEFAULT_NORMALIZATION = norm
                try:
                    per_crop = []
                    for cr in crops:
                        a, b = ds.crop(cr), dr.crop(cr)
                        per_crop.append(A.phase_cross_correlation(b, a, upsample_factor=100, normalization=norm)[0])
                        a.free(); b.free()
                    d0, f0 = A.align_image(ds, dr, use_autocorr=True, verbose=False)
                    # drift_diff_th = 0: no three crops agree exactly -> all eight are measured, then the closest three
                    d8, f8 = A.align_image(ds, dr, use_autocorr=True, drift_diff_th=0., verbose=False)
                    dref = A.DriftReference(dr)
                    try:
                        d1, f1 = A.align_image(ds, dref, use_autocorr=True, verbose=False)
                        d9, f9 = A.align_image(ds, dref, use_autocorr=True, drift_diff_th=0., verbose=False)
                    finally:
                        dref.free()
                finally:
                    A.DEFAULT_NORMALIZATION = old
                want, agree = A._consensus_drift(per_crop[:3], 3, 1.)
                if want is not None:
                    assert f0 == 0 and np.array_equal(d0, want), (dtype, norm, d0, want)
                assert f1 == f0 and np.array_equal(d1, d0), (dtype, norm)
                assert f8 == 1 and np.array_equal(d8, A._closest_three_drift(per_crop)), (dtype, norm, d8)
                assert f9 == 1 and np.array_equal(d9, d8), (dtype, norm)


def test_pairing_golden():
    from imageanalysis3_amd.spot_tools.matching import find_paired_centers, check_paired_centers
    g = load_golden("drift.npz")
    dr, pt, pr = find_paired_centers(g["pair_src_cts"], g["pair_ref_cts"], g["pair_rough"], cutoff=2.)
    assert np.array_equal(pt, g["pair_tar"]) and np.array_equal(pr, g["pair_ref"])
    assert np.allclose(dr, g["pair_drift"], atol=1e-12)
    if "check_drift" in g:
        dr2, pt2, pr2 = check_paired_centers(pt, pr, outlier_sigma=1.5)
        assert np.array_equal(pt2, g["check_tar"]) and np.allclose(dr2, g["check_drift"], atol=1e-12)


@pytest.mark.parametrize("norm", ["phase", None])
def test_phase_cross_correlation_vs_oracle_and_truth(norm):
    """Both normalisations against the oracle's restatement of the published algorithm and the analytically injected
    shift (normalization=None is additionally pinned against scikit-image 0.18.3 itself, see
    test_phase_correlation_and_align_image_vs_scikit_image_golden)."""
    import np_oracle as O
    from imageanalysis3_amd import synth
    from imageanalysis3_amd.correction_tools.alignment import phase_cross_correlation
    d = np.array([0.7, -3.25, 5.5])
    ref, src, _, _ = synth.make_bead_pair((20, 96, 96), 20, 3, d, margin=(5, 12, 12), min_sep=12.0)
    for up in (1, 10, 100):
        s, e, p = phase_cross_correlation(ref, src, upsample_factor=up, normalization=norm)
        so, eo, po = O.phase_cross_correlation(ref, src, upsample_factor=up, normalization=norm)
        assert np.allclose(s, so, atol=1e-9), (up, s, so)
        if up == 100:
            assert np.allclose(s, -d, atol=0.06)
    r16, s16 = ref.astype(np.uint16), src.astype(np.uint16)
    s, _, _ = phase_cross_correlation(r16, s16, upsample_factor=100, normalization=norm)
    so, _, _ = O.phase_cross_correlation(r16, s16, upsample_factor=100, normalization=norm)
    assert np.allclose(s, so, atol=1e-9)


def test_align_image_autocorr_known_answer(monkeypatch):
    from imageanalysis3_amd import synth
    from imageanalysis3_amd.correction_tools import alignment
    d = np.array([1.3, -4.6, 7.25])
    # un-normalised correlation (scikit-image < 0.19 behaviour): robust on small noisy crops
    ref, src, _, _ = synth.make_bead_pair((30, 256, 256), 120, 21, d)
    monkeypatch.setattr(alignment, "DEFAULT_NORMALIZATION", None)
    drift, flag = alignment.align_image(src, ref, use_autocorr=True, verbose=False)
    assert flag == 0
    assert np.allclose(drift, -d, atol=0.3), drift
    # (the phase-normalised variant is exercised at phase_cross_correlation level above: on 64x64
    #  crops holding ~7 beads each the crop-edge discontinuity dominates a phase-only spectrum, in the
    #  oracle exactly as on the device)
    with pytest.raises(IndexError):
        alignment.align_image(src, ref[:, :100], verbose=False)
    with pytest.raises(ValueError):
        alignment.align_image(src, ref, drift_channel='999', verbose=False)


# ---------------------------------------------------------------------------------------------
# warp
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["c1_f32", "c1_u16"])
def test_warp_golden_bit_exact(name):
    from imageanalysis3_amd.correction_tools.translate import warp_3d_image
    g = load_golden("warp.npz")
    im = build_case(name)[:, :96, :80]
    Z, X, Y = im.shape
    zz, xx, yy = np.meshgrid(np.arange(Z), np.arange(X), np.arange(Y), indexing="ij")
    field = np.stack([0.002 * (xx - X / 2), 0.01 * (yy - Y / 2) + 0.2, -0.008 * (xx - X / 2) + 0.005 * zz])
    for order, mode in ((1, "constant"), (3, "nearest"), (1, "nearest")):
        for use_field in (False, True):
            w = warp_3d_image(im, g["drift"], field if use_field else None, order, mode)
            key = "warp_%s_o%d_%s_f%d" % (name, order, mode, int(use_field))
            assert w.dtype == im.dtype and w.shape == im.shape
            ok = np.array_equal(w.reshape(-1)[g[key + "_idx"]], g[key + "_val"])
            assert ok and crc(w) == g[key + "_crc"], key


def test_warp_identity_and_integer_shift():
    import np_oracle as O
    from imageanalysis3_amd.correction_tools.translate import warp_3d_image
    im = build_case("edge_f32")
    assert np.array_equal(warp_3d_image(im, [0, 0, 0]), im)
    w = warp_3d_image(im, [1, -2, 3], warp_order=1, border_mode="nearest")
    assert np.array_equal(w, O.warp_3d_image(im, [1, -2, 3], None, 1, "nearest"))
    # large drifts: everything outside -> cval (constant) / edge (nearest, order 3)
    for order, mode in ((1, "constant"), (3, "nearest"), (3, "constant"), (0, "nearest"), (0, "constant")):
        w = warp_3d_image(im, [40.5, 100.25, -90.75], warp_order=order, border_mode=mode)
        assert np.array_equal(w, O.warp_3d_image(im, [40.5, 100.25, -90.75], None, order, mode))
    # order 0 (the nearest sample; the reference warps label images this way, segmentation_tools/cell.py:589): half-integer
    # coordinates, a field, both border modes, both dtypes
    rng = np.random.RandomState(2)
    lab = rng.randint(0, 40, size=(4, 30, 37)).astype(np.uint16)
    zz, xx, yy = np.meshgrid(np.arange(4), np.arange(30), np.arange(37), indexing="ij")
    fld = np.stack([0.5 * np.ones_like(zz), 1.5 * np.cos(yy / 5.0), 0.5 + 0.25 * xx]).astype(np.float32)
    for a in (lab, lab.astype(np.float32)):
        for mode in ("nearest", "constant"):
            for drift, f in (([0.5, -1.5, 2.5], None), ([0.25, 3.5, -7.75], fld), ([0, 0, 0], fld)):
                assert np.array_equal(warp_3d_image(a, drift, f, 0, mode), O.warp_3d_image(a, drift, f, 0, mode)), (mode, drift)
    with pytest.raises(NotImplementedError):
        warp_3d_image(im, [0, 0, 0], warp_order=2, border_mode="constant")


def _start_sum_cases():
    """Stacks whose long lines (> 566 padded samples: z^n underflows, the start sum of the prefilter is cut) begin with
    samples that contribute nothing, so that terms far down the line decide the first coefficient."""
    rng = np.random.RandomState(11)
    Z, X, Y = 5, 600, 580
    a = np.zeros((Z, X, Y), np.float32)                    # 150 zero rows / columns, then data
    a[:, 150:, 150:] = rng.uniform(100, 4000, size=(Z, X - 150, Y - 150))
    b = np.zeros((Z, X, Y), np.float32)                    # zeros, then values of 1e30: the tail dwarfs the head
    b[:, 70:, 66:] = rng.uniform(1e29, 1e30, size=(Z, X - 70, Y - 66))
    c = rng.uniform(1e-30, 1e-28, size=(Z, X, Y)).astype(np.float32)   # tiny head, huge tail
    c[:, 200:, 130:] = rng.uniform(1e20, 1e24, size=(Z, X - 200, Y - 130))
    d = np.zeros((Z, X, Y), np.uint16)                     # uint16 with a dark border
    d[:, 90:-40, 120:-64] = rng.randint(200, 60000, size=(Z, X - 130, Y - 184))
    e = np.zeros((Z, X, Y), np.float32)                    # nothing but zeros and one far sample per line
    e[:, -1, :] = 3.0
    e[:, :, -1] = 7.0
    return dict(zero_border=a, huge_tail=b, tiny_head=c, dark_border_u16=d, far_sample=e)


@pytest.mark.parametrize("name", ["zero_border", "huge_tail", "tiny_head", "dark_border_u16", "far_sample"])
def test_warp_cubic_start_sums_of_long_lines(name):
    """The causal start value of SciPy's spline prefilter sums the whole line; the kernels stop where the rest
    provably cannot change the sum (csrc/warp.hip, IirInit).  Lines that begin with zeros or with samples 50 orders of
    magnitude below their tail are where the sum has to read on (a fixed 64-term cut moves the float64 coefficients
    there by 1e-12 relative at the edge of the 12-sample pad; float32 outputs round that away almost always, which
    is not a proof): bit for bit against scipy."""
    import np_oracle as O
    from imageanalysis3_amd.correction_tools.translate import warp_3d_image
    im = _start_sum_cases()[name]
    drift = [0.3, 1.7, -2.2]
    w = warp_3d_image(im, drift, warp_order=3, border_mode="nearest")
    ref = O.warp_3d_image(im, drift, None, 3, "nearest")
    assert w.dtype == ref.dtype
    assert np.array_equal(w.view(np.uint32 if im.dtype == np.float32 else np.uint16),
                          ref.view(np.uint32 if im.dtype == np.float32 else np.uint16))


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
def test_warp_cubic_one_pass_prefilter_and_grouped_gather_bit_exact(dtype):
    """The spline prefilter along the long axes reads and writes every sample once: the anticausal recursion of a tile
    starts from a value certified by two bounding chains (csrc/warp.hip, spline_iir_strided_1p_k), lines whose chains do
    not meet fall back to two sweeps; the gather makes four outputs per thread from shared coefficient runs and sends
    outputs whose cells do not line up through per-tap loads.  IA3_TUNE_WARP_ONEPASS: default, warm-ups of 1, 3 and 9 samples
    (almost every / many / some tiles fail their certificate, at different places in a line) and of 32 (a mixture), two sweeps, and two
    sweeps with the one-output gather — all bit for bit equal to scipy, with a field whose components cross
    integers inside rows, a drift beyond the padding on one axis (runs leave the padded row), and stretches of zeros."""
    import np_oracle as O
    from imageanalysis3_amd import _lib as L
    from imageanalysis3_amd.correction_tools.translate import warp_3d_image
    rng = np.random.RandomState(5)
    Z, X, Y = 4, 300, 420
    im = rng.randint(90, 5000, size=(Z, X, Y)).astype(dtype)
    im[:, 40:44, :] = 0                    # short runs of zeros: the chains meet later there
    im[:, :, 200:260] = 0                  # a stretch longer than any warm-up
    im[1, 100:140, 300:380] = 60000
    zz, xx, yy = np.meshgrid(np.arange(Z), np.arange(X), np.arange(Y), indexing="ij")
    field = np.stack([0.4 * np.sin(yy / 17.0), 1.3 * np.cos(yy / 23.0) + 0.01 * xx, 0.9 * np.sin(yy / 9.0) + 0.004 * xx]).astype(np.float32)
    view = np.uint32 if dtype == np.float32 else np.uint16
    for drift, fld in (([0.3, 1.7, -2.2], field), ([0.3, 1.7, -2.2], None), ([-0.4, 2.5, 14.6], field),
                       ([0.2, -15.3, 0.7], field.astype(np.float64))):
        ref = O.warp_3d_image(im, drift, fld, 3, "nearest")
        try:
            for knob in (64, 1, 3, 9, 32, 0, -1):
                L.check(L.lib().ia3_set_tuning(12, knob))      # IA3_TUNE_WARP_ONEPASS
                w = warp_3d_image(im, drift, fld, 3, "nearest")
                assert w.dtype == ref.dtype
                assert np.array_equal(w.view(view), ref.view(view)), (drift, fld is not None, knob)
        finally:
            L.check(L.lib().ia3_set_tuning(12, 64))


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
def test_warp_cubic_axis0_pass_by_depth_bit_exact(dtype):
    """The spline prefilter's axis-0 pass keeps the padded z line in registers: depths of the build list (30) have an
    instantiation of their own, other depths from 8 to 64 planes (18, 41) get theirs from the run-time compiler
    (csrc/rtc.cpp), deeper stacks (70) take the generic kernels.  All bit for bit equal to scipy."""
    import np_oracle as O
    from imageanalysis3_amd.correction_tools.translate import warp_3d_image
    rng = np.random.RandomState(21)
    view = np.uint32 if dtype == np.float32 else np.uint16
    for Z in (30, 18, 41, 70):
        im = rng.randint(90, 5000, size=(Z, 24, 40)).astype(dtype)
        im[Z // 3] = 0
        ref = O.warp_3d_image(im, [0.4, -1.3, 2.6], None, 3, "nearest")
        w = warp_3d_image(im, [0.4, -1.3, 2.6], None, 3, "nearest")
        assert np.array_equal(w.view(view), ref.view(view)), Z


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
def test_warp_cubic_constant_mode_bit_exact(dtype):
    """warp_3d_image(warp_order=3) with its DEFAULT border mode 'constant' (correction_tools/translate.py:5-31): SciPy
    filters without padding and with the mirror boundary, returns cval = min(image) for coordinates outside the array
    and mirrors taps that leave it.  Bit for bit against scipy: drift alone, drift + field (both dtypes of field, both
    orders of applying them), a drift that pushes a third of the outputs outside, short axes (2 and 3 samples), an
    axis longer than the 566 samples after which the boundary sum's power underflows."""
    import np_oracle as O
    from imageanalysis3_amd import _lib as L
    from imageanalysis3_amd.correction_tools.translate import warp_3d_image
    rng = np.random.RandomState(11)
    view = np.uint32 if dtype == np.float32 else np.uint16
    for shape in ((5, 40, 70), (2, 3, 640), (3, 33, 20)):
        Z, X, Y = shape
        im = rng.randint(90, 5000, size=shape).astype(dtype)
        im[:, X // 2:X // 2 + 2, :] = 0
        zz, xx, yy = np.meshgrid(np.arange(Z), np.arange(X), np.arange(Y), indexing="ij")
        field = np.stack([0.4 * np.sin(yy / 17.0), 1.3 * np.cos(yy / 23.0) + 0.01 * xx, 0.9 * np.sin(yy / 9.0) + 0.004 * xx]).astype(np.float32)
        for drift, fld in (([0.3, 1.7, -2.2], None), ([0.3, 1.7, -2.2], field), ([-0.4, 0.5, 0.25 * Y], field.astype(np.float64)),
                           ([0.0, 0.0, 0.0], None), ([1e-9, -1e-9, 0.0], field)):
            ref = O.warp_3d_image(im, drift, fld, 3, "constant")
            w = warp_3d_image(im, drift, fld, 3, "constant")
            assert w.dtype == ref.dtype
            assert np.array_equal(w.view(view), ref.view(view)), (shape, drift, fld is not None)
    with pytest.raises(RuntimeError):
        warp_3d_image(np.zeros((1, 8, 8), dtype), [0, 0, 0], None, 3, "constant")


def test_stack_outlives_the_thread_that_made_it():
    """A host thread that ends returns its streams (csrc/runtime.cpp, ThreadCtx::~ThreadCtx).  Stacks it made stay valid:
    another thread reads them, frees them — ia3_stack_free would otherwise record an event on the dead thread's stream —
    and the blocks go back to the scratch cache; the cache's driver-call counters stay flat when the same work is repeated."""
    import ctypes as C
    import threading
    from imageanalysis3_amd import _lib as L
    from imageanalysis3_amd.correction_tools.filter import gaussian_high_pass_filter
    rng = np.random.RandomState(0)
    im = rng.randint(100, 4000, size=(6, 64, 80)).astype(np.uint16)
    want = gaussian_high_pass_filter(im, 3, 2)
    made = []

    def worker():
        src = L.DeviceStack.upload(im)
        out = L.DeviceStack.empty(im.shape, im.dtype)
        w, r = L.gaussian_taps(3, 2)
        L.check(L.lib().ia3_gaussian_highpass_dev(src._h, C.c_double(3), C.c_double(2), L.dptr(w), int(r), out._h))
        made.append((src, out))
    stats = (C.c_double * 6)()
    for rep in range(3):
        t = threading.Thread(target=worker)
        t.start(); t.join()                       # the thread (and its streams) are gone
        src, out = made.pop()
        assert np.array_equal(out.download(), want)
        src.free(); out.free()
        if rep == 1:
            L.check(L.lib().ia3_workspace_stats(stats))
            calls = (stats[3], stats[4])
    L.check(L.lib().ia3_workspace_stats(stats))
    assert (stats[3], stats[4]) == calls          # the third round took its blocks from the cache


def test_gaussianfit_class_vs_oracle():
    """(a3) standalone GaussianFit on explicit voxel lists (Voronoi cells of the golden case)."""
    import np_oracle as O
    from imageanalysis3_amd.External.Fitting_v4 import GaussianFit, gaussfit_batch
    for name in ("c1_f32", "c1_u16"):
        im = build_case(name)
        seeds = O.get_seeds(im, th_seed=600)
        f = O.iter_fit_seed_points(im, seeds.T)
        f.firstfit()
        gp = f.gparms[:12]
        ps, xs, ok, nfev = gaussfit_batch([g[0] for g in gp], [g[1] for g in gp], [g[2] for g in gp], delta_center=1.0)
        assert ok.all() and (nfev > 2).all()
        ref = np.array(f.ps[:12], dtype=np.float64)
        assert (np.abs(ps[:, :8] - ref[:, :8]) / np.abs(ref[:, :8])).max() <= 1e-4
        im_, X, c = gp[0]
        obj = GaussianFit(im_, X, center=c, delta_center=1.0)
        p0 = obj.p.copy()
        oo = O.GaussianFit(im_, X, center=c, delta_center=1.0)
        assert np.array_equal(obj.p_, oo.p_) and np.allclose(p0, oo.p, rtol=1e-6)   # same start point
        obj.fit()
        assert obj.success and np.array_equal(obj.p, ps[0])
        oo.fit()
        assert np.allclose(obj.get_im(), oo.get_im(), rtol=1e-4, atol=1e-6)
    # < 10 voxels: fit refuses (Fitting_v4.py:382-383)
    small = GaussianFit(np.arange(8, dtype=np.float32), np.zeros((3, 8), dtype=int), center=[0, 0, 0])
    small.fit()
    assert small.success is False


# ---------------------------------------------------------------------------------------------
# (a13) elementwise pre-corrections
# ---------------------------------------------------------------------------------------------
def _profiles(X, Y, dtype):
    xx, yy = np.meshgrid(np.arange(X), np.arange(Y), indexing="ij")
    illum = np.exp(-(((xx - X / 2) / (0.9 * X)) ** 2 + ((yy - Y / 2) / (0.8 * Y)) ** 2)).astype(dtype)
    illum /= illum.max()
    bleed = np.zeros((3, 3, X, Y), dtype=dtype)
    for i in range(3):
        for j in range(3):
            bleed[i, j] = (1.0 if i == j else -0.05 * (1 + 0.1 * i + 0.03 * j)) * (1 + 0.02 * illum)
    return illum, bleed


@pytest.mark.parametrize("pdtype", [np.float32, np.float64])
def test_illumination_and_bleedthrough_bit_exact(pdtype):
    import np_oracle as O
    from imageanalysis3_amd.io_tools.load import illumination_correction, bleedthrough_correction
    ims = [build_case("c1_u16"), build_case("hot_u16"), (build_case("c1_u16")[::-1] // 2 + 7).astype(np.uint16)]
    ims[2][3, 5, 7] = 65535  # forces the upper clip in the mix
    illum, bleed = _profiles(128, 128, pdtype)
    for im in ims[:2]:
        assert np.array_equal(illumination_correction(im, illum), O.illumination_correction(im, illum))
    got = bleedthrough_correction(ims, bleed)
    ref = O.bleedthrough_correction(ims, bleed)
    for g_, r_ in zip(got, ref):
        assert g_.dtype == np.uint16 and np.array_equal(g_, r_)


@pytest.mark.parametrize("name", ["c1_u16", "hot_u16", "c1_f32"])
def test_z_shift_correction_bit_exact(name):
    import np_oracle as O
    from imageanalysis3_amd.corrections import Z_Shift_Correction
    im = build_case(name)
    if name == "c1_u16":
        im = (im * (1 + 0.01 * np.arange(im.shape[0]))[:, None, None]).astype(np.uint16)  # real z trend
    assert np.array_equal(Z_Shift_Correction(im), O.z_shift_correction(im))
    odd = np.ascontiguousarray(im[:, :17, :19])   # odd plane size: single middle element
    assert np.array_equal(Z_Shift_Correction(odd), O.z_shift_correction(odd))


# ---------------------------------------------------------------------------------------------
# full size (BASELINE.json configs[1]): size-independent properties
# ---------------------------------------------------------------------------------------------
def test_full_size_fov_properties():
    """2048x2048x50 float32, 5000 isolated spots: every injected spot is called once and recovered (centre,
    height, widths, background), the run is deterministic, and a crop of the field seeded/fitted on its own
    gives the same rows for the spots whose neighbourhood lies inside the crop (shift consistency)."""
    from imageanalysis3_amd import synth
    from imageanalysis3_amd.spot_tools.fitting import fit_fov_image
    from scipy.spatial import cKDTree
    shape = (50, 2048, 2048)
    im, c, h = synth.make_fov(shape, 5000, 3)
    t = fit_fov_image(im, "647", th_seed=600, max_num_seeds=None, verbose=False)
    assert t.shape == (5000, 11) and t.dtype == np.float32
    d, j = cKDTree(c).query(t[:, 1:4])
    assert d.max() < 0.35 and len(np.unique(j)) == 5000                      # one row per injected spot
    assert np.median(d) < 0.03
    assert np.median(np.abs(t[:, 0] / h[j] - 1)) < 0.02                        # heights
    assert np.abs(np.median(t[:, 4]) - 400) < 1.0                              # background
    assert np.allclose(np.median(t[:, 5:8], axis=0), [1.35, 1.9, 1.9], atol=0.02)
    assert (np.diff(fit_fov_image(im, "647", th_seed=600, max_num_seeds=None, verbose=False), axis=0) ==
            np.diff(t, axis=0)).all()                                          # deterministic
    # shift consistency: same spots from a crop (spots >= 40 px from the crop border see identical filters
    # up to the 30-px reflect halo; their fits depend on a 5-px ball only)
    x0, y0, n = 512, 768, 512
    sub = np.ascontiguousarray(im[:, x0:x0 + n, y0:y0 + n])
    ts = fit_fov_image(sub, "647", th_seed=600, max_num_seeds=None, verbose=False)
    inner = (ts[:, 2] > 40) & (ts[:, 2] < n - 40) & (ts[:, 3] > 40) & (ts[:, 3] < n - 40)
    ts = ts[inner]
    ts[:, 2] += x0
    ts[:, 3] += y0
    d, j = cKDTree(t[:, 1:4]).query(ts[:, 1:4])
    assert len(ts) > 150 and d.max() < 1e-3
    rel = np.abs(ts[:, :8].astype(np.float64) - t[j, :8]) / np.abs(t[j, :8])
    assert rel.max() < 1e-4


# ---- (a12) legacy per-cell path: visual_tools.get_seed_in_distance + Fitting_v3 + _fit_single_image --------
def _tie_canon(s):
    s = np.asarray(s)
    return s[np.lexsort((s[:, 2], s[:, 1], s[:, 0], -s[:, 3]))] if len(s) else s.reshape(0, 4)


def test_legacy_seed_in_distance_golden():
    from conftest import build_legacy
    from imageanalysis3_amd import visual_tools as vt
    im, m = build_legacy()
    g = load_golden("legacy.npz")
    for name, sa in m["seeding"].items():
        for i, cc in enumerate(g["coords"]):
            got = vt.get_seed_in_distance(im, cc, *sa)
            ref = g["seeds_%s_%d" % (name, i)]
            assert got.dtype == np.int64 and got.shape == ref.shape, (name, i, got.shape, ref.shape)
            assert np.array_equal(got[:, 3], ref[:, 3]), (name, i)          # brightest first, same heights
            assert np.array_equal(_tie_canon(got), _tie_canon(ref)), (name, i)
    got = vt.get_seed_in_distance(im, None, 0, 30, 0.75, 10, 3, True, 95, 300, True, 10, 2, 1, 4, True)
    assert np.array_equal(_tie_canon(got), _tie_canon(g["seeds_whole_per"]))
    assert np.array_equal(vt.get_seed_points_base(im, 0.75, 5, 3, 500, 2, True), g["base_bg5"])
    with pytest.raises(ValueError):
        vt.get_seed_in_distance(im, [1, 2])
    # float32 stack: integer-valued maxima only (the int64 cast of the reference), same as the oracle
    import np_oracle as O
    imf = im.astype(np.float32)
    a = vt.get_seed_in_distance(imf, g["coords"][0], 0, 30, 0.75, 10, 3, False, 95, 300, True, 10, 2, 1, 4, True)
    b = O.legacy_get_seed_in_distance(imf, g["coords"][0], 0, 30, 0.75, 10, 3, False, 95, 300, True, 10, 2, 1, 4, True)
    assert np.array_equal(_tie_canon(a), _tie_canon(b))


def test_legacy_fit_single_image_golden():
    from conftest import build_legacy
    from imageanalysis3_amd.classes import _fit_single_image
    from imageanalysis3_amd.External import Fitting_v3
    im, m = build_legacy()
    g = load_golden("legacy.npz")
    sa = tuple(m["seeding"]["default"][:-1]) + (False,)
    fa = tuple(m["fitting_args"])
    out = _fit_single_image(im, 0, g["coords"], sa, fa)
    assert len(out) == len(g["coords"])
    for i, sp in enumerate(out):
        ref = g["fit_%d" % i]
        if len(ref) == 0:
            assert len(sp) == 0
            continue
        assert sp.shape == ref.shape
        np.testing.assert_allclose(sp[:, :8], ref[:, :8], rtol=RTOL, atol=1e-4)
        np.testing.assert_allclose(sp[:, 8:], ref[:, 8:], rtol=1e-3, atol=2e-3)
    # first fit alone + sweep count
    from imageanalysis3_amd import visual_tools as vt
    s = vt.get_seed_in_distance(im, g["coords"][0], *sa)
    f = Fitting_v3.iter_fit_seed_points(im, s.T, *fa)
    f.firstfit()
    np.testing.assert_allclose(np.array(f.ps)[:, :8], g["first_0"][:, :8], rtol=RTOL, atol=1e-4)
    f.repeatfit()
    assert f.n_iter == int(g["n_iter_0"])
    with pytest.raises(ValueError):
        Fitting_v3.iter_fit_seed_points(im, np.zeros((3, 0)), *fa).firstfit()
    with pytest.raises(NotImplementedError):
        Fitting_v3.iter_fit_seed_points(im, s.T, *fa, weight_sigma=1.)


# ---- background level / intensity normalisation (background.hip) -------------------------------------------
def test_find_image_background_golden_bit_exact():
    from conftest import special_background_images
    from imageanalysis3_amd.io_tools.load import find_image_background
    g = load_golden("norm.npz")
    for k, im in special_background_images().items():
        assert find_image_background(im) == g["special_" + k], k
        assert find_image_background(im, max_iter=1) == g["special_i1_" + k], k
    for name in ["c1_u16", "c1_f32", "hot_u16", "edge_f32"]:
        im = build_case(name)
        assert find_image_background(im) == g[name + "_back"], name
        assert find_image_background(im, bin_size=25, max_iter=3) == g[name + "_back_b25_i3"], name
    # non-uniform edges go through the binary search: same answer as NumPy on the oracle
    import np_oracle as O
    im = build_case("c1_u16")
    assert find_image_background(im, bin_size=7) == O.find_image_background(im, bin_size=7)
    assert find_image_background(im[3], bin_size=10) == O.find_image_background(im[3], bin_size=10)   # 2-D input


@pytest.mark.parametrize("name", ["c1_u16", "c1_f32", "hot_u16", "edge_f32"])
def test_local_background_and_normalised_rows_golden(name):
    from imageanalysis3_amd import _lib as L
    from imageanalysis3_amd.io_tools.load import find_local_backgrounds
    from imageanalysis3_amd.spot_tools.fitting import fit_fov_image
    g = load_golden("norm.npz")
    im = build_case(name)
    st = L.DeviceStack.upload(im)
    try:
        backs = find_local_backgrounds(st, g[name + "_plain"][:, 1:4], 10)
    finally:
        st.free()
    assert np.array_equal(backs, g[name + "_backs"])            # integer histogram work: bit-exact
    for key, kw in (("_local", dict(normalize_local=True)), ("_global", dict(normalize_background=True))):
        rows = fit_fov_image(im, "647", th_seed=600, verbose=False, **kw)
        ref = g[name + key]
        ia, ib = match_rows(rows, ref)
        np.testing.assert_allclose(rows[ia][:, :8], ref[ib][:, :8], rtol=RTOL, atol=1e-4)


# ---- (f1) correct_fov_image: the whole pre-correction chain on resident uint16 stacks ----------------------------
@pytest.mark.parametrize("name", ["full", "silent_no_warp", "highpass", "no_drift_647_only", "no_hot_f64_illum"])
def test_correct_fov_image_chain_golden_bit_exact(name, tmp_path):
    from conftest import build_chain_case, chain_kwargs, write_dax
    from imageanalysis3_amd.io_tools.load import correct_fov_image
    case = build_chain_case()
    g = load_golden("chain.npz")
    sel, kw = chain_kwargs(case, name)
    path = str(tmp_path / "movie.dax")
    write_dax(path, case["raw"])
    out = correct_fov_image(path, sel, **kw)
    assert isinstance(out, tuple) and len(out) == 1
    for ch, im in zip(sel, out[0]):
        ref = g["%s_%s" % (name, ch)]
        assert im.dtype == np.uint16 and im.shape == ref.shape
        assert np.array_equal(im, ref), (name, ch, int((im != ref).sum()))
    # raw movie passed directly, results left resident and fed to the fitter without another upload
    out2 = correct_fov_image(case["raw"], sel, return_device=True, return_drift=True, **kw)
    stacks, drift, flag = out2
    try:
        assert flag == 0 and np.allclose(drift, kw["drift"] if kw["drift"] is not None else 0)
        for ch, st in zip(sel, stacks):
            assert np.array_equal(st.download(), g["%s_%s" % (name, ch)])
        if name == "full":
            from imageanalysis3_amd.spot_tools.fitting import fit_fov_image
            a = fit_fov_image(stacks[0], sel[0], th_seed=300, normalize_local=True, verbose=False)
            b = fit_fov_image(stacks[0].download(), sel[0], th_seed=300, normalize_local=True, verbose=False)
            assert len(a) > 0 and np.array_equal(a, b)
    finally:
        for st in stacks:
            st.free()


def test_correct_fov_image_argument_errors(tmp_path):
    from conftest import build_chain_case, chain_kwargs
    from imageanalysis3_amd.io_tools.load import correct_fov_image, split_im_by_channels, read_dax
    case = build_chain_case()
    sel, kw = chain_kwargs(case, "full")
    with pytest.raises(IOError):
        correct_fov_image(str(tmp_path / "missing.dax"), sel, **kw)
    bad = dict(kw); bad["drift_channel"] = '405'
    with pytest.raises(ValueError):
        correct_fov_image(case["raw"], sel, **bad)
    bad = dict(kw); bad["illumination_profile"] = {'750': case["illum"]['750']}
    with pytest.raises(KeyError):
        correct_fov_image(case["raw"], sel, **bad)
    bad = dict(kw); bad["drift"] = [1., 2.]
    with pytest.raises(IndexError):
        correct_fov_image(case["raw"], sel, **bad)
    with pytest.raises(NotImplementedError):
        correct_fov_image(case["raw"], sel, normalization=True, **kw)
    # host split == device split
    a = split_im_by_channels(case["raw"], ['561', '488'], case["chs"], [case["Z"], case["X"], case["Y"]], case["nb"], 0)
    from imageanalysis3_amd import _lib as L
    st = L.DeviceStack.upload(case["raw"])
    try:
        b = split_im_by_channels(st, ['561', '488'], case["chs"], [case["Z"], case["X"], case["Y"]], case["nb"], 0)
        for x, y in zip(a, b):
            assert np.array_equal(x, y.download())
            y.free()
    finally:
        st.free()


def test_correct_fov_image_translation_functions():
    """warp_image=False (io_tools/load.py:454-486): images stay unwarped, one spot-translation function per channel."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    from make_golden import chromfn_inputs
    from conftest import build_chain_case, chain_kwargs
    from imageanalysis3_amd.io_tools.load import correct_fov_image
    case = build_chain_case()
    g = load_golden("chain.npz")
    gf = load_golden("chromfn.npz")
    info, coords, spots, drift = chromfn_inputs()
    sel, kw = chain_kwargs(case, "silent_no_warp")
    kw["chromatic_profile"] = {'750': info, '647': None, '561': info}
    kw["drift"] = drift
    ims, funcs = correct_fov_image(case["raw"], sel, warp_image=False, **kw)
    for ch, im in zip(sel, ims):
        assert np.array_equal(im, g["silent_no_warp_%s" % ch])       # unwarped images
    assert len(funcs) == 2
    assert np.array_equal(funcs[0](spots), gf["spots"])               # 750: chromatic constants + drift
    assert np.array_equal(funcs[1](spots), gf["drift_only"])          # 647 (reference channel): drift only


# ---- the step driver's operator sequence (classes/preprocess.py:337-1260), replayed flat ---------------------------
@pytest.mark.parametrize("tag,rescale,illum64", [("a", True, False), ("b", False, True)])
def test_daxprocesser_steps_golden_bit_exact(tag, rescale, illum64, tmp_path):
    """The fixtures are the reference's step driver run on a synthetic movie; what they pin is the operator sequence with
    that driver's arithmetic (tests/harness/replay.py): every step bit-exact (CRC + sampled voxels)."""
    from conftest import build_chain_case, write_dax
    from harness import replay as R
    from imageanalysis3_amd.spot_tools.fitting import fit_fov_image
    from imageanalysis3_amd.correction_tools.alignment import align_image
    case = build_chain_case()
    g = load_golden("daxp.npz")
    chs = case["chs"]
    path = str(tmp_path / "movie.dax")
    write_dax(path, case["raw"])
    st = R.load_channels(path, chs, chs, [case["Z"], case["X"], case["Y"]], n_buffer=case["nb"])

    def check(step, channels):
        for c in channels:
            im = st[c].download()
            key = "%s_%s_%s" % (tag, step, c)
            idx, vals = g[key + "_smp"]
            assert im.dtype == np.uint16
            assert np.array_equal(im.reshape(-1)[idx], vals.astype(np.uint16)), (key, "sampled voxels differ")
            assert crc(im) == g[key + "_crc"], key

    try:
        R.hot_pixels_in_image_dtype(st, chs)
        check("hot", chs)
        R.bleedthrough_rescaled(st, chs[:3], case["bleed"], rescale)     # the bead channel is not part of the mix
        check("bleed", chs[:3])
        illum = {k: (a.astype(np.float64) if illum64 else a) for k, a in case["illum"].items()}
        R.illumination_rescaled(st, chs, illum, rescale)
        check("illum", chs)
        # the signal channels get drift + chromatic field (none for the reference channel), the bead channel the drift only
        R.warp_drift_then_field(st, chs, np.array(case["drift"]), {c: case["chrom"].get(c) for c in chs[:3]})
        check("warp", chs)
        if tag == "a":
            R.highpass(st, ['750'])
            check("highpass", ['750'])
            for c in ('647', '561'):
                ref = g["a_spots_%s" % c]
                got = fit_fov_image(st[c], c, th_seed=300, max_num_seeds=None, verbose=False)
                # seeds sit 3-4 voxels apart here: exact Voronoi ties exist; the library resolves them by cKDTree's layout
                # as the reference does.
                import np_oracle as O
                orc = O.fit_fov_image(st[c].download(), c, th_seed=300, max_num_seeds=None)
                ia, ib = match_rows(got, orc)
                rel = np.abs(got[ia][:, :8] - orc[ib][:, :8]) / np.maximum(np.abs(orc[ib][:, :8]), 1e-3)
                # one fit of this noisy rescaled field runs into maxfev = 1000 without converging (MINPACK warns about it in
                # the oracle too); where such a fit stops depends on the last bits of every step, and the seeds coupled to it
                # through the refit sweeps inherit the difference: that group is held to 3e-2 (the non-converged row itself
                # moves by ~1 % between MINPACK's QR and the kernel's Cholesky with Newton-refined pivots), the rest to 1e-5
                assert (rel.max(1) > 1e-5).sum() <= 3 and rel.max() < 3e-2, rel.max(1)
                ia, ib = match_rows(got, ref)
                np.testing.assert_allclose(got[ia][:, :8], ref[ib][:, :8], rtol=1e-2, atol=1e-3)
            # drift of the bead channel against itself through align_image: zeros, flag 0
            d, f = align_image(st['488'], st['488'], drift_channel='488', all_channels=chs, verbose=False)
            assert f == 0 and np.abs(d).max() == 0
    finally:
        R.free_all(st)


def test_fit_fov_images_concurrent_equals_sequential():
    """Several host threads, one stream each: same tables as one call after the other."""
    from imageanalysis3_amd.spot_tools.fitting import fit_fov_image, fit_fov_images
    ims = [build_case(n) for n in ("c1_u16", "c1_f32", "m_f32", "hot_u16", "clu_f32", "edge_f32")] * 2
    seq = [fit_fov_image(im, "647", th_seed=600, verbose=False) for im in ims]
    for workers in (2, 4):
        par = fit_fov_images(ims, ["647"] * len(ims), n_workers=workers, th_seed=600)
        assert len(par) == len(seq)
        for a, b in zip(par, seq):
            assert a.shape == b.shape and np.array_equal(a, b)


def test_group_fit_of_several_fields_equals_one_by_one():
    """ia3_fit_fovs fits a group of images with ONE fitter (the work list runs over the seeds of every field, so the
    dependent chains and maxfev fits of one field overlap with the others' work): neighbours, Voronoi ties and sweep
    order never cross an image, so every table equals the one ia3_fit_fov_dev makes — isolated, crowded (exact ties:
    the per-field seed trees) and uint16 fields of one shape mixed in one batch, more images than one group."""
    import np_oracle as O
    from imageanalysis3_amd import synth, _lib as L
    shape = (30, 160, 160)
    ims = []
    for k in range(7):
        layout = "clustered" if k % 2 else "isolated"
        im, c, h = synth.make_fov(shape, 60 if layout == "isolated" else 90, 20 + k, layout=layout, n_territories=5)
        ims.append(im)
    ims.append(np.zeros(shape, np.float32) + 400)          # a field without seeds
    sp, keep = L.make_seed_params(600.0, max_num_seeds=None)
    fp = L.make_fit_params()
    one = [L.fit_fovs([im], sp, fp, in_flight=1) for im in ims]
    assert sum(len(t[0][0]) for t in one) > 300
    for dtype in (np.float32, np.uint16):
        batch = [im.astype(dtype) for im in ims] if dtype != np.float32 else ims
        ref = one if dtype == np.float32 else [L.fit_fovs([im], sp, fp, in_flight=1) for im in batch]
        for depth in (3, 8):
            tabs, info = L.fit_fovs(batch, sp, fp, in_flight=depth)
            for k, (t, i) in enumerate(zip(tabs, info)):
                assert np.array_equal(t, ref[k][0][0]), (dtype, depth, k)
                for key in ("n_seeds", "n_iter", "fits", "nfev", "voxel_evals"):
                    assert i[key] == ref[k][1][0][key], (dtype, depth, k, key)
    # and against the oracle (the reference's cKDTree rule on the crowded ones)
    o = O.fit_fov_image(ims[1], "647", th_seed=600, max_num_seeds=None)
    assert_rows_close(one[1][0][0], o, rtol=1e-6)


def test_fit_fovs_waits_for_stacks_still_in_production_on_the_callers_stream():
    """A resident stack handed to ia3_fit_fovs may still be in production on the caller's stream (the correction chain
    is asynchronous); the batch seeds on streams of its own.  Here the stack holds zeros until a copy that is queued
    behind some 50 ms of other work on the caller's stream: seeding that does not order itself after the caller's
    queue sees the zeros (no seeds).  No ia3_sync() anywhere before the call."""
    import ctypes as C
    from imageanalysis3_amd import synth, _lib as L
    lib = L.lib()
    shape = (30, 200, 200)
    im, c, h = synth.make_fov(shape, 80, 31)
    sp, keep = L.make_seed_params(600.0, max_num_seeds=None)
    fp = L.make_fit_params()
    ref = L.fit_fovs([im, im], sp, fp, in_flight=1)[0]
    assert len(ref[0]) > 60
    big = np.random.RandomState(2).uniform(100, 900, size=(40, 1024, 1024)).astype(np.float32)
    zero = np.zeros(3)
    with L.DeviceStack.upload(big) as a, L.DeviceStack.empty(big.shape, big.dtype) as b, \
            L.DeviceStack.upload(im) as src, L.DeviceStack.upload(np.zeros(shape, np.float32)) as t1, \
            L.DeviceStack.upload(np.zeros(shape, np.float32)) as t2:
        L.check(lib.ia3_sync())
        for _ in range(12):   # cubic warps of a 168 MB stack: a few ms each
            L.check(lib.ia3_warp3d_dev(a._h, L.dptr(np.array([0.3, 0.4, 0.5])), None, 0, 3, L.MODE_NEAREST, C.c_double(0.0), b._h))
        for t in (t1, t2):    # zero shift, order 1: a copy
            L.check(lib.ia3_warp3d_dev(src._h, L.dptr(zero), None, 0, 1, L.MODE_NEAREST, C.c_double(0.0), t._h))
        tabs, info = L.fit_fovs([t1, t2], sp, fp, in_flight=2)
    for t in tabs:
        assert np.array_equal(t, ref[0])


def _set_gauss_cert(v):
    import ctypes as C
    from imageanalysis3_amd import _lib as L
    L.check(L.lib().ia3_set_tuning(C.c_int(1), C.c_int(v)))


def test_gaussian_long_filter_certified_fused_path_bit_exact():
    """The sigma-7.5 passes use fused multiply-adds where the float32 / uint16 value is provably unaffected and the
    reference operation sequence elsewhere.  Same bits as SciPy with the default guard, with the fused path off and
    with every output sent through the reference sequence, on inputs built to sit on quantisation boundaries:
    piecewise-constant uint16 (every sum is within an ulp of an integer, truncation flips on the last bit),
    zeros, mixed signs and negative zeros (fused path must step aside)."""
    from scipy import ndimage as ndi
    from imageanalysis3_amd.correction_tools.filter import gaussian_filter
    rng = np.random.RandomState(11)
    shape = (40, 256, 384)
    blocks = np.zeros(shape, np.uint16)
    vals = rng.randint(1, 65535, size=(2, 3))
    for i in range(2):
        for j in range(3):
            blocks[:, i * 128:(i + 1) * 128, j * 128:(j + 1) * 128] = vals[i, j]
    pos = rng.gamma(2.0, 300.0, size=shape).astype(np.float32)
    pos[:, :64, :64] = 0                      # exact zeros
    pos[:, 64:128, :64] = 400.0               # constant block
    mixed = rng.normal(0, 300.0, size=shape).astype(np.float32)
    negz = pos.copy(); negz[5:9, 100:140, 200:260] = -0.0
    u16 = np.clip(rng.gamma(2.0, 300.0, size=shape), 0, 65535).astype(np.uint16)
    cases = {"blocks_u16": blocks, "pos_f32": pos, "mixed_f32": mixed, "negzero_f32": negz, "gamma_u16": u16}
    try:
        for name, im in cases.items():
            ref = ndi.gaussian_filter(im, 7.5, mode="reflect", truncate=4.0)
            for cert in (-2, -1, 1 << 28):
                _set_gauss_cert(cert)
                got = gaussian_filter(im, 7.5, mode="reflect", truncate=4.0)
                if im.dtype == np.float32:   # compare bit patterns: -0.0 vs +0.0 counts
                    same = np.array_equal(got.view(np.uint32), ref.view(np.uint32))
                else:
                    same = np.array_equal(got, ref)
                assert same, (name, cert, int((got != ref).sum()))
    finally:
        _set_gauss_cert(-2)


def test_gaussian_axis0_folded_column_pass_bit_exact():
    """Stacks of the built depths (Makefile FOLD_DEPTHS: 25 30 33 35 40 45 50 60) and of a depth compiled at run time (48) run the axis-0 pass of a long filter with
    the whole column in registers and the border folded into the weights (IA3_TUNE_GAUSS_FOLD, a different summation order, certified like the fused path).
    Same bits as SciPy for both border modes, with the default guard, with every output sent through the reference
    sequence, with the guard forced wide open and with the folded form off, on inputs that sit on quantisation
    boundaries (see the test above)."""
    import ctypes as C
    from scipy import ndimage as ndi
    from imageanalysis3_amd import _lib as L
    from imageanalysis3_amd.correction_tools.filter import gaussian_filter
    rng = np.random.RandomState(12)
    # a depth without a translation unit of its own: its kernels come from the run-time compiler (hiprtc, the same kernel
    # text and flags; csrc/gauss_col_dispatch.hip) — ~35 s per dtype the first time on a machine, then from the cache
    for dt in (L.dtype_code(np.zeros(1, np.float32)), L.dtype_code(np.zeros(1, np.uint16))):
        assert L.lib().ia3_prepare_depth(dt, 48) == 2
        assert L.lib().ia3_prepare_depth(dt, 50) == 1
    assert L.lib().ia3_prepare_depth(L.dtype_code(np.zeros(1, np.float32)), 70) == 0      # beyond 64 planes: window kernels
    try:
        for Z in (25, 30, 33, 35, 40, 45, 50, 60, 48):
            shape = (Z, 96, 192)
            blocks = np.zeros(shape, np.uint16)
            for i in range(2):
                for j in range(3):
                    blocks[:, i * 48:(i + 1) * 48, j * 64:(j + 1) * 64] = rng.randint(1, 65535)
            blocks[Z // 2:, :, 100:] //= 3                      # a step along the filtered axis as well
            pos = rng.gamma(2.0, 300.0, size=shape).astype(np.float32)
            pos[:, :32, :32] = 0
            pos[:, 32:64, :32] = 400.0
            mixed = rng.normal(0, 300.0, size=shape).astype(np.float32)
            negz = pos.copy(); negz[3:6, 40:60, 100:130] = -0.0
            u16 = np.clip(rng.gamma(2.0, 300.0, size=shape), 0, 65535).astype(np.uint16)
            cases = {"blocks_u16": blocks, "pos_f32": pos, "mixed_f32": mixed, "negzero_f32": negz, "gamma_u16": u16}
            for name, im in cases.items():
                for mode in ("reflect", "nearest"):
                    ref = ndi.gaussian_filter(im, 7.5, mode=mode, truncate=4.0)
                    for fold, cert in ((1, -2), (1, -1), (1, 1 << 28), (0, -2)):
                        L.check(L.lib().ia3_set_tuning(C.c_int(8), C.c_int(fold)))
                        _set_gauss_cert(cert)
                        got = gaussian_filter(im, 7.5, mode=mode, truncate=4.0)
                        if im.dtype == np.float32:
                            same = np.array_equal(got.view(np.uint32), ref.view(np.uint32))
                        else:
                            same = np.array_equal(got, ref)
                        assert same, (Z, name, mode, fold, cert, int((got != ref).sum()))
    finally:
        _set_gauss_cert(-2)
        L.check(L.lib().ia3_set_tuning(C.c_int(8), C.c_int(1)))


def test_dog_filter_pair_shared_axis0_launch_bit_exact():
    """ia3_dog_filters_dev: the seed detector's two filtered stacks.  On the built depths (25 ... 60) the two axis-0 passes come
    from one launch (column in registers) and the short filter's other axes from the plane-wise kernel — so does a depth whose kernels are compiled at run time (48); depths outside 16 ... 64 (12)
    run the separate filters.  front == scipy gaussian_filter(im, 0.75), back == gaussian_filter1d(im, 7.5, axis=0),
    bit for bit, on ragged plane sizes, with the guard at its default, off and wide open."""
    import ctypes as C
    from scipy import ndimage as ndi
    from imageanalysis3_amd import _lib as L
    lib = L.lib()
    rng = np.random.RandomState(13)
    try:
        for Z, X, Y in ((50, 150, 530), (30, 64, 248), (40, 37, 90), (50, 16, 8), (48, 80, 120), (12, 40, 64), (45, 20, 96), (25, 40, 64),
                        (33, 33, 128), (35, 48, 200), (60, 24, 160)):
            shape = (Z, X, Y)
            pos = rng.gamma(2.0, 300.0, size=shape).astype(np.float32)
            pos[:, :20, :20] = 0
            mixed = rng.normal(0, 300.0, size=shape).astype(np.float32)
            u16 = np.clip(rng.gamma(2.0, 300.0, size=shape), 0, 65535).astype(np.uint16)
            steps = np.zeros(shape, np.uint16)
            steps[:, X // 2:, :] = 40000; steps[Z // 3:, :, Y // 2:] += 7777
            for name, im in (("pos_f32", pos), ("mixed_f32", mixed), ("gamma_u16", u16), ("steps_u16", steps)):
                ref_f = ndi.gaussian_filter(im, 0.75, mode="reflect", truncate=4.0)
                ref_b = ndi.gaussian_filter1d(im, 7.5, axis=0, mode="reflect", truncate=4.0)
                for cert in (-2, -1, 1 << 28):
                    _set_gauss_cert(cert)
                    with L.DeviceStack.upload(im) as st, L.DeviceStack.empty(shape, im.dtype) as f, \
                            L.DeviceStack.empty(shape, im.dtype) as b:
                        L.check(lib.ia3_dog_filters_dev(st._h, C.c_double(0.75), C.c_double(7.5), f._h, b._h))
                        got_f, got_b = f.download(), b.download()
                    v = np.uint32 if im.dtype == np.float32 else np.uint16
                    assert np.array_equal(got_f.view(v), ref_f.view(v)), (shape, name, cert, "front", int((got_f != ref_f).sum()))
                    assert np.array_equal(got_b.view(v), ref_b.view(v)), (shape, name, cert, "back", int((got_b != ref_b).sum()))
    finally:
        _set_gauss_cert(-2)


# ---------------------------------------------------------------------------------------------
# production entry: movie -> corrected images + drift + spots in the FOV save file
# ---------------------------------------------------------------------------------------------
def _spot_tables_close(got, ref, what):
    """Stored (n, L, 11) tables: same occupied rows, rows within the fit tolerance (order = fit order)."""
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    occ_g, occ_r = got.any(axis=2), ref.any(axis=2)
    assert np.array_equal(occ_g, occ_r), (what, occ_g.sum(1), occ_r.sum(1))
    for i in range(len(ref)):
        n = int(occ_r[i].sum())
        if n:
            a, b = got[i, :n].astype(np.float64), ref[i, :n].astype(np.float64)
            # background column of spots on removed hot columns is exp(-large): compare it on an absolute scale
            cols = [0, 1, 2, 3, 5, 6, 7]
            rel = np.abs(a[:, cols] - b[:, cols]) / np.abs(b[:, cols])
            assert rel.max() <= 1e-3, (what, i, rel.max())
            assert (rel <= RTOL).mean() >= 0.9, (what, i, (rel <= RTOL).mean())
            assert np.abs(a[:, 4] - b[:, 4]).max() <= 1e-3 * max(np.abs(b[:, 4]).max(), 1.0), (what, i)


@pytest.mark.parametrize("tag,warp", [("w_", True), ("n_", False)])
def test_batch_process_image_to_spots_golden(tag, warp, tmp_path):
    """classes/batch_functions.py:60-303 run by the reference (h5py) on the synthetic movie: first pass with a stored
    drift, second pass resuming from the save file (images carried over, tables kept), third pass overwriting the
    spots.  Images, flags and drifts identical; spot tables within the fit tolerance."""
    from conftest import batch_inputs, write_dax
    from imageanalysis3_amd.classes import batch_functions as B
    from imageanalysis3_amd.io_tools import h5lite as H
    if not H.available():
        pytest.skip("libhdf5 not present")
    gold = load_golden("h5batch.npz")
    case, size, corr, corr_nowarp, fit = batch_inputs()
    os.makedirs(str(tmp_path / "H1R1"))
    movie = str(tmp_path / "H1R1" / "Conv_zscan_05.dax")
    write_dax(movie, case["raw"])
    ref_im = np.zeros(size, np.uint16)
    cargs = corr if warp else corr_nowarp
    path = str(tmp_path / "fov.hdf5")
    B.create_fov_save_file(path, 'unique', [5, 2, 9], ['750', '647', '561'], size, max_num_seeds=4)
    with H.File(path, "a", libver="latest") as f:
        f['unique']['drifts'][:2, :] = np.array(case["drift"], np.float32)

    def check(prefix):
        with H.File(path, "r") as f:
            g = f['unique']
            for k in ('ids', 'channels', 'flags', 'drifts'):
                assert np.array_equal(g[k][...], gold[prefix + k]), (prefix, k)
            crcs = [zlib.crc32(np.ascontiguousarray(g['ims'][i]).tobytes()) & 0xFFFFFFFF for i in range(3)]
            assert crcs == [int(c) for c in gold[prefix + 'ims_crc']], (prefix, crcs)
            _spot_tables_close(g['spots'][...], gold[prefix + 'spots'], prefix + 'spots')
            _spot_tables_close(g['raw_spots'][...], gold[prefix + 'raw_spots'], prefix + 'raw_spots')

    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        out = B.batch_process_image_to_spots(movie, ['750', '647'], path, 'unique', [5, 2], ref_im, warp_image=warp,
                                             correction_args=dict(cargs), fitting_args=dict(fit), verbose=warp)
        assert out is None
        check(tag)
        B.batch_process_image_to_spots(movie, ['750', '647'], path, 'unique', [5, 2], ref_im, warp_image=warp,
                                       correction_args=dict(cargs), fitting_args=dict(fit, max_num_seeds=3), verbose=warp)
        check(tag + "again_")
        sp, raw = B.batch_process_image_to_spots(movie, ['750', '647'], path, 'unique', [5, 2], ref_im, warp_image=warp,
                                                 correction_args=dict(cargs), fitting_args=dict(fit, max_num_seeds=3),
                                                 overwrite_spot=True, verbose=warp, return_spots=True)
        check(tag + "over_")
    assert len(sp) == 2 and all(len(s) <= 3 for s in sp)
    # the channels of a movie fitted one after the other or by inner threads: same tables
    one = B.batch_process_image_to_spots(movie, ['750', '647'], path, 'unique', [5, 2], ref_im, warp_image=warp, save_spots=False,
                                         correction_args=dict(cargs), fitting_args=dict(fit), return_spots=True, fit_workers=1)
    two = B.batch_process_image_to_spots(movie, ['750', '647'], path, 'unique', [5, 2], ref_im, warp_image=warp, save_spots=False,
                                         correction_args=dict(cargs), fitting_args=dict(fit), return_spots=True, fit_workers=2)
    for a, b in zip(one[0] + one[1], two[0] + two[1]):
        assert np.array_equal(a, b)
    # argument checks of the reference (:92-118)
    with pytest.raises(IOError):
        B.batch_process_image_to_spots(movie[:-4] + ".tif", ['750'], path, 'unique', [5], ref_im)
    with pytest.raises(IOError):
        B.batch_process_image_to_spots(movie, ['750'], path[:-5] + ".h5", 'unique', [5], ref_im)
    with pytest.raises(TypeError):
        B.batch_process_image_to_spots(movie, ['750'], path, 'unique', [5], 3)
    with pytest.raises(ValueError):
        B.batch_process_image_to_spots(movie, ['750', '647'], path, 'unique', [5], ref_im)


def test_daxprocesser_fit_spots_by_segmentation_golden(tmp_path):
    """classes/preprocess.py:1093-1153 against the reference's own run, as a flat replay: per-label crops (with and
    without a drift) cut on the device, fit_fov_image on each, kept spots and their labels; a label whose box holds no
    seed contributes nothing."""
    from conftest import seg_labels, build_chain_case, write_dax
    from harness import replay as R
    g = load_golden("seg.npz")
    case = build_chain_case()
    size = [case["Z"], case["X"], case["Y"]]
    lab = seg_labels(size)
    assert (zlib.crc32(np.ascontiguousarray(lab).tobytes()) & 0xFFFFFFFF) == int(g["lab_crc"])
    zz, xx, yy = np.where(lab == 2)
    assert R.label_box(lab == 2).tolist() == [[max(zz.min() - 1, 0), min(zz.max() + 2, size[0])], [xx.min() - 1, xx.max() + 2],
                                              [yy.min() - 1, yy.max() + 2]]
    path = str(tmp_path / "movie.dax")
    write_dax(path, case["raw"])
    st = R.load_channels(path, case["chs"], case["chs"], size, n_buffer=case["nb"])
    try:
        R.hot_pixels_in_image_dtype(st, case["chs"])
        s647, i647 = R.fit_in_labels(st['647'], '647', lab, np.zeros(3), th_seed=300, search_radius=3)
        drift = np.array(case["drift"])
        s750, i750 = R.fit_in_labels(st['750'], '750', lab, drift, th_seed=300, num_spots=2)
        s561, i561 = R.fit_in_labels(st['561'], '561', (lab == 4) * 4, drift, th_seed=300)
    finally:
        R.free_all(st)
    assert np.array_equal(i647, g["ids_647"]) and i647.dtype == np.int32
    assert_rows_close(np.asarray(s647), g["spots_647"])
    assert np.array_equal(i750, g["ids_750"])
    assert_rows_close(np.asarray(s750), g["spots_750"])
    assert len(s561) == 0 and len(i561) == 0 and g["spots_561"].shape == (0,)


def test_profiles_read_from_correction_folder(tmp_path):
    """Profiles left as None are read from the correction folder under the reference's file names
    (io_tools/load.py:239-281, :553-640): same images as with the profiles handed over; the step driver's operator
    sequence with profiles read the same way reproduces its fixture."""
    import contextlib, io
    from conftest import build_chain_case, chain_kwargs, write_dax
    from imageanalysis3_amd.io_tools.load import correct_fov_image, load_correction_profile
    from harness import replay as R
    case = build_chain_case()
    g = load_golden("chain.npz")
    Z, X, Y = case["Z"], case["X"], case["Y"]
    chs = case["chs"]
    folder = str(tmp_path / "corr")
    os.makedirs(folder)
    for c in chs:
        np.save(os.path.join(folder, "illumination_correction_%s_%dx%d.npy" % (c, X, Y)), case["illum"][c])
    np.save(os.path.join(folder, "bleedthrough_correction_750_647_561_%d_%d.npy" % (X, Y)), case["bleed"].reshape(-1))
    for c in ('750', '561'):
        np.save(os.path.join(folder, "chromatic_correction_%s_647_%d_%d_%d.npy" % (c, Z, X, Y)), case["chrom"][c])
    path = str(tmp_path / "movie.dax")
    write_dax(path, case["raw"])
    sel, kw = chain_kwargs(case, "full")
    for k in ("illumination_profile", "bleed_profile", "chromatic_profile"):
        kw.pop(k)
    with contextlib.redirect_stdout(io.StringIO()):
        out = correct_fov_image(path, sel, correction_folder=folder, **kw)
    for ch, im in zip(sel, out[0]):
        assert np.array_equal(im, g["full_%s" % ch]), ch
    with pytest.raises(FileNotFoundError):
        correct_fov_image(path, sel, correction_folder=str(tmp_path), **kw)
    # the step driver's sequence with every profile read from the folder
    d = load_golden("daxp.npz")
    common = dict(correction_folder=folder, all_channels=chs, im_size=[Z, X, Y])
    bleed = load_correction_profile('bleedthrough', chs[:3], ref_channel=chs[0], **common)
    illum = load_correction_profile('illumination', chs, ref_channel=chs[0], **common)
    chrom = load_correction_profile('chromatic', chs[:3], ref_channel='647', **common)
    st = R.load_channels(path, chs, chs, [Z, X, Y], n_buffer=case["nb"])
    try:
        R.hot_pixels_in_image_dtype(st, chs)
        R.bleedthrough_rescaled(st, chs[:3], bleed, True)
        R.illumination_rescaled(st, chs, illum, True)
        R.warp_drift_then_field(st, chs, np.array(case["drift"]), chrom)
        for c in chs:
            assert (zlib.crc32(np.ascontiguousarray(st[c].download()).tobytes()) & 0xFFFFFFFF) == int(d["a_warp_%s_crc" % c]), c
    finally:
        R.free_all(st)


def test_batch_process_images_to_spots_threads_equal_sequential(tmp_path, monkeypatch):
    """Several movies of one FOV into one save file, three ways: one after the other (the per-movie entry), through the
    pipelined library call (ia3_process_movies: upload | corrections + drift + warps | cross-movie group fits) and as a
    pool of host threads — same file content every time (and the reference's single-movie fixture for the ids they
    share)."""
    import contextlib, io
    from conftest import batch_inputs, write_dax
    from imageanalysis3_amd.classes import batch_functions as B
    from imageanalysis3_amd.io_tools import h5lite as H
    if not H.available():
        pytest.skip("libhdf5 not present")
    case, size, corr, corr_nowarp, fit = batch_inputs()
    ref_im = np.zeros(size, np.uint16)
    movies, ids = [], []
    for r in range(4):   # four "rounds": the same frames under different names, two regions each
        os.makedirs(str(tmp_path / ("H%dR%d" % (r, r))))
        m = str(tmp_path / ("H%dR%d" % (r, r)) / "Conv_zscan_05.dax")
        write_dax(m, case["raw"])
        movies.append(m)
        ids.append([10 + 2 * r, 11 + 2 * r])
    all_ids = [i for pair in ids for i in pair]

    from imageanalysis3_amd import _lib as L
    calls = []
    real = L.process_movies
    monkeypatch.setattr(L, "process_movies", lambda *a, **k: (calls.append(len(a[1])), real(*a, **k))[1])

    def run(path, threads, stored_drift=True, ref=ref_im, pipeline=None):
        B.create_fov_save_file(path, 'unique', all_ids, ['750', '647'] * 4, size, max_num_seeds=4)
        if stored_drift:
            with H.File(path, "a", libver="latest") as f:
                f['unique']['drifts'][...] = np.array(case["drift"], np.float32)
        args = [dict(dax_filename=m, sel_channels=['750', '647'], region_ids=i) for m, i in zip(movies, ids)]
        shared = dict(save_filename=path, data_type='unique', ref_filename=ref, warp_image=True,
                      correction_args=dict(corr), fitting_args=dict(fit), verbose=True)
        with contextlib.redirect_stdout(io.StringIO()):
            out = B.batch_process_images_to_spots(args, num_threads=threads, shared_kwargs=shared, pipeline=pipeline)
        assert out == [None] * 4
        with H.File(path, "r") as f:
            return {k: f['unique'][k][...] for k in ('ims', 'spots', 'raw_spots', 'flags', 'drifts')}

    seq = run(str(tmp_path / "seq.hdf5"), 1)
    assert calls == []
    par = run(str(tmp_path / "par.hdf5"), 4)
    assert calls == [4]                                   # all four movies went through ONE pipelined call
    thr = run(str(tmp_path / "thr.hdf5"), 4, pipeline=False)
    assert calls == [4]
    for k in seq:
        assert np.array_equal(seq[k], par[k]), k
        assert np.array_equal(seq[k], thr[k]), k
    assert (seq['flags'] == 2).all() and seq['spots'].any(axis=(1, 2)).all()
    # drift measured per movie (phase correlation of the bead channel against a reference bead image): every thread
    # runs its own FFT plans, results do not depend on the number of threads
    nb, Z = case["nb"], case["Z"]
    bead = np.ascontiguousarray(case["raw"][nb + (3 - nb) % 4::4][:Z])
    bead_ref = np.roll(bead, (1, -2), axis=(1, 2))
    seq_d = run(str(tmp_path / "seq_d.hdf5"), 1, stored_drift=False, ref=bead_ref)
    par_d = run(str(tmp_path / "par_d.hdf5"), 4, stored_drift=False, ref=bead_ref)
    thr_d = run(str(tmp_path / "thr_d.hdf5"), 4, stored_drift=False, ref=bead_ref, pipeline=False)
    assert calls == [4, 4]
    for k in seq_d:
        assert np.array_equal(seq_d[k], par_d[k]), k
        assert np.array_equal(seq_d[k], thr_d[k]), k
    # a save file that already holds these movies' images: nothing for the pipelined entry to do, the per-movie rules apply
    again = run(str(tmp_path / "par_d.hdf5"), 4, stored_drift=False, ref=bead_ref)
    assert calls == [4, 4]
    for k in seq_d:
        assert np.array_equal(seq_d[k], again[k]), k
    assert np.abs(seq_d['drifts']).max() > 0.5 and (seq_d['drifts'] == seq_d['drifts'][0]).all()
    gold = load_golden("h5batch.npz")
    for r in range(4):   # every round holds the frames of the single-movie fixture
        assert (zlib.crc32(np.ascontiguousarray(par['ims'][2 * r]).tobytes()) & 0xFFFFFFFF) == int(gold['w_ims_crc'][0])
        _spot_tables_close(par['spots'][2 * r:2 * r + 2], gold['w_spots'][:2], "round %d" % r)


@pytest.mark.parametrize("variant", ["full", "silent_no_warp", "highpass", "no_drift_647_only", "no_hot_f64_illum"])
def test_movie_pipeline_options_equal_the_per_movie_calls(variant):
    """ia3_process_movies (io_tools.load.MoviePlan) against correct_fov_image + fit_fov_image called movie by movie, over
    the option sets of the chain fixtures (z shift, silent call = no warp, high-pass without bleedthrough / chromatic
    correction, a single reference channel without drift, float64 illumination without hot-pixel removal) and over the
    fit options the plan carries (seed cap, global and local background normalisation): corrected images, drifts and
    tables identical; movies from host arrays, two per call; then the drift measured against a bead image."""
    import contextlib, io
    from conftest import build_chain_case, chain_kwargs
    from imageanalysis3_amd.io_tools.load import correct_fov_image, MoviePlan
    from imageanalysis3_amd.spot_tools.fitting import fit_fov_image
    case = build_chain_case()
    sel, kw = chain_kwargs(case, variant)
    raw = case["raw"]
    drift = kw.pop("drift")
    th = {c: 300. for c in sel}
    with contextlib.redirect_stdout(io.StringIO()):
        ref_ims, ref_drift, ref_flag = correct_fov_image(raw, sel, drift=drift, return_drift=True, **kw)
    for fit_kw in (dict(max_num_seeds=None), dict(max_num_seeds=20, normalize_local=True), dict(normalize_background=True)):
        ref_tabs = [fit_fov_image(im, c, th_seed=300, verbose=False, **fit_kw) for im, c in zip(ref_ims, sel)]
        plan = MoviePlan(sel, calculate_drift=False, seed_th=th, fitting_args=dict(fit_kw), frames=raw.shape[0], **kw)
        out = plan.run([raw, raw.copy()], drifts_in=[drift, drift], measure_drift=False, want_images=True)
        assert len(out) == 2
        for o in out:
            assert o["drift_flag"] == ref_flag
            assert np.array_equal(o["drift"], np.asarray(ref_drift, dtype=np.float64))
            for a, b in zip(o["images"], ref_ims):
                assert np.array_equal(a, b)
            for t, r, ns in zip(o["tables"], ref_tabs, o["n_seeds"]):
                if ns == 0:
                    assert len(r) == 0
                else:
                    assert t.shape == r.shape and np.array_equal(t, r)
        plan2 = MoviePlan(sel, calculate_drift=False, seed_th=th, fitting_args=dict(fit_kw), frames=raw.shape[0], fit_spots=False, **kw)
        o2 = plan2.run([raw], drifts_in=[drift], measure_drift=False, want_images=True)[0]
        assert all(np.array_equal(a, b) for a, b in zip(o2["images"], ref_ims)) and all(len(t) == 0 for t in o2["tables"])
    if variant != "full":
        return
    # measured drift: the bead channel against a moved copy of itself
    nb, Z = case["nb"], case["Z"]
    bead = np.ascontiguousarray(raw[nb + (3 - nb) % 4::4][:Z])
    bead_ref = np.roll(bead, (1, -2), axis=(1, 2))
    with contextlib.redirect_stdout(io.StringIO()):
        ref_ims, ref_drift, ref_flag = correct_fov_image(raw, sel, calculate_drift=True, ref_filename=bead_ref, return_drift=True, **kw)
    plan = MoviePlan(sel, ref_image=bead_ref, calculate_drift=True, seed_th=th, fitting_args=dict(max_num_seeds=None),
                     frames=raw.shape[0], **kw)
    o = plan.run([raw], want_images=True)[0]
    assert o["drift_flag"] == ref_flag and np.array_equal(o["drift"], ref_drift)
    assert np.abs(o["drift"]).max() > 0.5
    for a, b in zip(o["images"], ref_ims):
        assert np.array_equal(a, b)


def test_movie_pipeline_failures_are_reported_not_hidden(tmp_path):
    """ia3_process_movies / MoviePlan when things are wrong: a movie of another shape is refused before anything runs; a
    .dax file that is too short fails ITS movie with the library's message while the batch still drains (no thread left
    waiting); options the pipelined entry does not carry raise NotImplementedError at plan time (the caller then goes
    movie by movie); a row buffer that is too small comes back as a capacity error and the wrapper retries with room."""
    import contextlib, io
    from conftest import build_chain_case, chain_kwargs
    from imageanalysis3_amd import _lib as L
    from imageanalysis3_amd.io_tools.load import MoviePlan
    case = build_chain_case()
    sel, kw = chain_kwargs(case, "full")
    raw = case["raw"]
    drift = kw.pop("drift")
    th = {c: 300. for c in sel}
    plan = MoviePlan(sel, calculate_drift=False, seed_th=th, fitting_args=dict(max_num_seeds=None), frames=raw.shape[0], **kw)
    with pytest.raises(TypeError):
        plan.run([raw[:-4]], drifts_in=[drift], measure_drift=False)
    with pytest.raises(TypeError):
        plan.run([raw.astype(np.float32)], drifts_in=[drift], measure_drift=False)
    # a truncated movie file among good ones
    good = str(tmp_path / "good.dax")
    bad = str(tmp_path / "bad.dax")
    raw.tofile(good)
    raw[: raw.shape[0] // 2].tofile(bad)
    for name in (good, bad):
        with open(name[:-4] + ".inf", "w") as f:
            f.write("frame dimensions = %d x %d\nnumber of frames = %d\n" % (raw.shape[2], raw.shape[1], raw.shape[0]))
    ok = plan.run([good], drifts_in=[drift], measure_drift=False)
    with pytest.raises(ValueError) as ei:
        plan.run([good, bad, good], drifts_in=[drift] * 3, measure_drift=False)
    assert "movie 1" in str(ei.value) and "fewer than" in str(ei.value)
    again = plan.run([good, raw], drifts_in=[drift, drift], measure_drift=False)      # the library is still usable
    for t, r in zip(again[1]["tables"], ok[0]["tables"]):
        assert np.array_equal(t, r)
    # a row buffer too small for the tables: capacity error inside, retried by the wrapper
    small = plan.run([raw], drifts_in=[drift], measure_drift=False, capacity=4)
    for t, r in zip(small[0]["tables"], ok[0]["tables"]):
        assert np.array_equal(t, r)
    for bad_kw in (dict(warp_image=False), dict(normalization=True), dict(output_dtype=np.float32)):
        with pytest.raises(NotImplementedError):
            MoviePlan(sel, calculate_drift=False, seed_th=th, frames=raw.shape[0], **dict(kw, **bad_kw))


def test_movie_file_streams_into_a_resident_stack(tmp_path):
    """ia3_stack_load_file: pieces larger and smaller than the staging buffers, an offset, big-endian files; equal to
    read_dax + upload."""
    from conftest import write_dax
    from imageanalysis3_amd import _lib as L
    from imageanalysis3_amd.io_tools.load import read_dax, load_dax_resident
    rng = np.random.RandomState(8)
    for frames, X, Y in ((7, 33, 65), (40, 1024, 512)):      # 30 KB and 40 MiB (more than one 32 MiB piece)
        raw = rng.randint(0, 65535, size=(frames, X, Y)).astype(np.uint16)
        path = str(tmp_path / ("m%d.dax" % frames))
        write_dax(path, raw)
        st = load_dax_resident(path)
        try:
            assert st.shape == (frames, X, Y) and st.dtype == np.uint16
            assert np.array_equal(st.download(), raw) and np.array_equal(read_dax(path), raw)
        finally:
            st.free()
        part = L.DeviceStack.from_file(path, 3, X, Y, offset_bytes=2 * X * Y * 2)
        assert np.array_equal(part.download(), raw[2:5])
        part.free()
    big = str(tmp_path / "big.dax")
    raw.astype('>u2').tofile(big)
    with open(big[:-4] + ".inf", "w") as f:
        f.write("frame dimensions = %d x %d\nnumber of frames = %d\n big endian\n" % (Y, X, frames))
    st = load_dax_resident(big)
    assert np.array_equal(st.download(), raw) and np.array_equal(read_dax(big), raw)
    st.free()
    with pytest.raises(ValueError):
        L.DeviceStack.from_file(path, frames + 1, X, Y)
    with pytest.raises(ValueError):
        L.DeviceStack.from_file(str(tmp_path / "absent.dax"), 1, 8, 8)


def test_stack_cache_reuse_and_release():
    """Stacks come from the stream-ordered cache: a freed block is handed out again, contents of live stacks are never
    touched, ia3_release_workspace gives idle blocks back."""
    from imageanalysis3_amd import _lib as L
    rng = np.random.RandomState(1)
    a = rng.randint(0, 60000, size=(5, 64, 96)).astype(np.uint16)
    s1 = L.DeviceStack.upload(a)
    s2 = L.DeviceStack.upload(a + 1)
    s1.free()
    s3 = L.DeviceStack.upload(a + 2)          # may reuse s1's block; s2 must be intact
    assert np.array_equal(s2.download(), a + 1) and np.array_equal(s3.download(), a + 2)
    c = s3.crop([[1, 4], [10, 50], [20, 90]])
    assert np.array_equal(c.download(), (a + 2)[1:4, 10:50, 20:90])
    for s in (s2, s3, c):
        s.free()
    L.check(L.lib().ia3_release_workspace())
    s4 = L.DeviceStack.upload(a)
    assert np.array_equal(s4.download(), a)
    s4.free()


def test_phase_correlation_and_align_image_vs_scikit_image_golden(monkeypatch):
    """Pinned against the real thing: skimage.registration.phase_cross_correlation 0.18.3 (un-normalised correlation)
    and the reference's align_image on its default phase-correlation path, both run under /opt/conda's interpreter
    (oracle/make_golden_h5.py phase)."""
    from imageanalysis3_amd import synth
    from imageanalysis3_amd.correction_tools import alignment
    g = load_golden("phase.npz")
    dd = np.array([0.7, -3.25, 5.5])
    ref, src, _, _ = synth.make_bead_pair((20, 96, 96), 20, 3, dd, margin=(5, 12, 12), min_sep=12.0)
    for tag, a, b in (("f32", ref, src), ("u16", ref.astype(np.uint16), src.astype(np.uint16))):
        for up in (1, 10, 100):
            s, e, p = alignment.phase_cross_correlation(a, b, upsample_factor=up, normalization=None)
            exp = g["pcc_%s_%d" % (tag, up)]
            assert np.allclose(s, exp[:3], atol=1e-9), (tag, up, s, exp[:3])
            assert abs(e - exp[3]) <= 1e-5 and abs(p - exp[4]) <= 1e-5, (tag, up, e, p, exp[3:])
    d2 = np.array([1.3, -4.6, 7.25])
    ref2, src2, _, _ = synth.make_bead_pair((30, 256, 256), 120, 21, d2)
    crops = alignment.generate_drift_crops(single_im_size=[30, 256, 256])
    assert np.array_equal(np.asarray(crops), g["align_crops"])
    monkeypatch.setattr(alignment, "DEFAULT_NORMALIZATION", None)   # what scikit-image 0.17 / 0.18 compute
    for a, b, key in ((src2, ref2, "align"), (src2.astype(np.uint16), ref2.astype(np.uint16), "align_u16")):
        drift, flag = alignment.align_image(a, b, crop_list=crops, use_autocorr=True, verbose=False)
        assert flag == int(g[key + "_flag"])
        assert np.allclose(drift, g[key + "_drift"], atol=1e-9), (key, drift, g[key + "_drift"])


def test_batch_process_image_to_spots_measured_drift_golden(tmp_path):
    """The same entry with nothing stored: the drift comes from the phase correlation of the bead channel against a
    reference bead image (scikit-image in the reference's run), the images are warped with it, spots fitted — stored
    drift, flags and images identical to the reference's file, tables within the fit tolerance."""
    import contextlib, io
    from conftest import batch_inputs, write_dax
    from imageanalysis3_amd.classes import batch_functions as B
    from imageanalysis3_amd.io_tools import h5lite as H
    if not H.available():
        pytest.skip("libhdf5 not present")
    gold = load_golden("h5batch.npz")
    case, size, corr, corr_nowarp, fit = batch_inputs()
    os.makedirs(str(tmp_path / "H1R1"))
    movie = str(tmp_path / "H1R1" / "Conv_zscan_05.dax")
    write_dax(movie, case["raw"])
    nb, Z = case["nb"], case["Z"]
    bead = np.ascontiguousarray(case["raw"][nb + (3 - nb) % 4::4][:Z])
    bead_ref = np.roll(bead, (1, -2), axis=(1, 2))
    path = str(tmp_path / "fov.hdf5")
    B.create_fov_save_file(path, 'unique', [5, 2, 9], ['750', '647', '561'], size, max_num_seeds=4)
    with contextlib.redirect_stdout(io.StringIO()):
        B.batch_process_image_to_spots(movie, ['750', '647'], path, 'unique', [5, 2], bead_ref, warp_image=True,
                                       correction_args=dict(corr), fitting_args=dict(fit), verbose=True)
    with H.File(path, "r") as f:
        g = f['unique']
        assert np.array_equal(g['flags'][...], gold['d_flags'])
        assert np.allclose(g['drifts'][...], gold['d_drifts'], atol=1e-6), (g['drifts'][...], gold['d_drifts'])
        crcs = [zlib.crc32(np.ascontiguousarray(g['ims'][i]).tobytes()) & 0xFFFFFFFF for i in range(3)]
        assert crcs == [int(c) for c in gold['d_ims_crc']], crcs
        _spot_tables_close(g['spots'][...], gold['d_spots'], 'd_spots')
        _spot_tables_close(g['raw_spots'][...], gold['d_raw_spots'], 'd_raw_spots')


def test_upsampled_dft_on_matrix_cores_equals_vector_unit():
    """The three contractions of the upsampled DFT (skimage's _upsampled_dft) run as v_mfma_f64 tiles; the first
    version on the vector unit is kept behind IA3_TUNE_DFT_VALU: same shifts, error and phase to rounding, on crops
    whose sizes are not multiples of the tiles."""
    import ctypes as C
    from imageanalysis3_amd import synth, _lib as L
    from imageanalysis3_amd.correction_tools import alignment
    out = {}
    cases = []
    for shape, nb, d in (((20, 96, 96), 20, (0.7, -3.25, 5.5)), ((13, 70, 121), 12, (-1.31, 2.77, -0.46))):
        ref, src, _, _ = synth.make_bead_pair(shape, nb, 3, np.array(d), margin=(3, 10, 10), min_sep=10.0)
        cases += [(ref, src), (ref.astype(np.uint16), src.astype(np.uint16))]
    try:
        for valu in (0, 1):
            L.check(L.lib().ia3_set_tuning(C.c_int(2), C.c_int(valu)))
            out[valu] = [alignment.phase_cross_correlation(a, b, upsample_factor=up, normalization=nm)
                         for a, b in cases for up in (10, 100) for nm in (None, "phase")]
    finally:
        L.check(L.lib().ia3_set_tuning(C.c_int(2), C.c_int(0)))
    for (s0, e0, p0), (s1, e1, p1) in zip(out[0], out[1]):
        assert np.array_equal(s0, s1), (s0, s1)
        assert abs(e0 - e1) <= 1e-9 and abs(p0 - p1) <= 1e-9
