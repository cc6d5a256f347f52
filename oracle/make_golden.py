"""TEST INFRASTRUCTURE — generate tests/golden/*.npz by running the REFERENCE's own Python.

Run only in the development container (needs /root/reference):

    python oracle/make_golden.py

Inputs are never stored: every fixture records the generator call
(``imageanalysis3_amd.synth``; bit-stable integer-hash generator) and the reference outputs.
The reference is loaded file-by-file via ``oracle/ref_loader.py``.
"""
import os
import sys
import io
import json
import zlib
import contextlib
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_loader  # noqa: E402
from imageanalysis3_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def samples(a, n=4096, seed=99):
    idx = (synth.uniform01(seed, 11, np.arange(n)) * a.size).astype(np.int64)
    return idx, a.reshape(-1)[idx].copy()


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def case_image(spec):
    """spec -> stack; mirrored by tests/conftest.py::build_case."""
    im, c, h = synth.make_fov(tuple(spec["shape"]), spec["n"], spec["seed"], layout=spec["layout"],
                              dtype=np.dtype(spec["dtype"]),
                              **{k: spec[k] for k in ("n_territories", "min_sep", "margin") if k in spec})
    if spec.get("hot_columns"):
        for (x, y, v) in spec["hot_columns"]:
            im[:, x, y] = v
    return im


CASES = {
    "c1_f32": dict(shape=[30, 128, 128], n=50, seed=1, layout="isolated", dtype="float32"),
    "c1_u16": dict(shape=[30, 128, 128], n=50, seed=1, layout="isolated", dtype="uint16"),
    "m_f32": dict(shape=[50, 256, 256], n=150, seed=2, layout="isolated", dtype="float32"),
    "edge_f32": dict(shape=[12, 40, 56], n=8, seed=4, layout="isolated", dtype="float32", min_sep=10.0,
                     margin=[1, 3, 3]),
    "clu_f32": dict(shape=[30, 128, 128], n=80, seed=5, layout="clustered", dtype="float32",
                    n_territories=6),
    "hot_u16": dict(shape=[30, 128, 128], n=30, seed=6, layout="isolated", dtype="uint16",
                    hot_columns=[[40, 41, 9000], [90, 17, 12000]]),
    # layout B of SURVEY.md §8(d) at 50 x 256 x 256: 16 territories of ~15 spots, exact Voronoi ties in most of them
    "club_f32": dict(shape=[50, 256, 256], n=240, seed=8, layout="clustered", dtype="float32",
                     n_territories=16),
}


LEGACY = dict(shape=[30, 160, 160], n=14, seed=11, layout="clustered", dtype="uint16", margin=[5, 12, 12],
              n_territories=2)
# (_seeding_args, label) tuples: positional arguments after (im, center) of get_seed_in_distance
LEGACY_SEEDING = {
    "default": (0, 30, 0.75, 10, 3, False, 95, 300, True, 10, 2, 1, 4, True),
    "top3": (3, 30, 0.75, 10, 3, False, 95, 300, True, 10, 2, 1, 4, True),
    "static": (0, 30, 0.75, 10, 3, False, 95, 300, False, 10, 2, 1, 4, True),
    "r12_th2000": (0, 12, 0.75, 10, 3, False, 95, 2000, True, 10, 2, 1, 4, True),
    "many_dynamic": (0, 30, 0.75, 10, 3, False, 95, 6000, True, 10, 8, 1, 4, True),
}
LEGACY_FIT_ARGS = (5, 1., 2.5, 10, 0.1)


def legacy_golden(meta):
    """(a12) classes/__init__.py:57-88: the reference's visual_tools.get_seed_in_distance and
    Fitting_v3.iter_fit_seed_points run in the order _fit_single_image calls them."""
    F3, vt = ref_loader.load_legacy()
    spec = dict(LEGACY)
    im, c, h = synth.make_fov(tuple(spec["shape"]), spec["n"], spec["seed"], layout=spec["layout"],
                              dtype=np.dtype(spec["dtype"]), margin=tuple(spec["margin"]),
                              n_territories=spec["n_territories"])
    coords = np.array([c[:7].mean(0), c[7:].mean(0), [3., 150., 20.], [15., 20., 60.]])
    d = {"coords": coords, "crc": crc(im)}
    for name, sa in LEGACY_SEEDING.items():
        for i, cc in enumerate(coords):
            d["seeds_%s_%d" % (name, i)] = quiet(vt.get_seed_in_distance, im, cc, *sa)
    d["seeds_whole_per"] = quiet(vt.get_seed_in_distance, im, None, 0, 30, 0.75, 10, 3, True, 95, 300, True, 10, 2, 1, 4, True)
    d["base_bg5"] = quiet(vt.get_seed_points_base, im, 0.75, 5, 3, 500, 2, True)
    sa = LEGACY_SEEDING["default"][:-1] + (False,)
    norm = np.nanmedian(im)
    for i, cc in enumerate(coords):
        s = quiet(vt.get_seed_in_distance, im, cc, *sa)
        if len(s) == 0:
            d["fit_%d" % i] = np.zeros((0, 11), np.float32)
            continue
        f = F3.iter_fit_seed_points(im, s.T, *LEGACY_FIT_ARGS)
        quiet(f.firstfit)
        d["first_%d" % i] = np.array(f.ps, dtype=np.float32)
        quiet(f.repeatfit)
        sp = np.array(f.ps)
        sp[:, 0] = sp[:, 0] / norm
        d["fit_%d" % i] = sp
        d["n_iter_%d" % i] = f.n_iter
    np.savez_compressed(os.path.join(OUT, "legacy.npz"), **d)
    meta["legacy"] = {"image": LEGACY, "seeding": {k: list(v) for k, v in LEGACY_SEEDING.items()},
                      "fitting_args": list(LEGACY_FIT_ARGS)}


NORM_CASES = ["c1_u16", "c1_f32", "hot_u16", "edge_f32"]


def special_background_images():
    """Small stacks that drive find_image_background (io_tools/load.py:642-687) through its corners."""
    rng = np.random.RandomState(4)
    d = {}
    d["const"] = np.full((6, 12, 12), 500, np.uint16)                      # single bin > edge rule -> still a peak
    d["const0"] = np.zeros((6, 12, 12), np.uint16)                         # everything in bin 0: no peak -> median
    d["plateau"] = np.repeat(np.array([100, 110, 120, 130], np.uint16), 216).reshape(6, 12, 12)   # flat top
    d["two_peaks_tie"] = np.repeat(np.array([100, 300, 500, 700], np.uint16), [300, 132, 300, 132]).reshape(6, 12, 12)
    d["top_edge"] = np.full((6, 12, 12), 65530, np.uint16)                 # == last edge: closed last bin
    d["above_range"] = np.full((6, 12, 12), 65534, np.uint16)              # dropped by the histogram
    x = rng.normal(420., 35., size=(8, 20, 20))
    d["noise_u16"] = np.clip(x, 0, 65535).astype(np.uint16)
    d["noise_f32"] = x.astype(np.float32)
    y = x.astype(np.float32).copy(); y[0, :3, :3] = np.nan
    d["nan_f32"] = y
    d["sparse"] = (rng.randint(0, 6000, size=(4, 8, 8)) * 10).astype(np.uint16)   # every count 0 or 1-2
    return d


def norm_golden(meta):
    """Background normalisation of fit_fov_image (spot_tools/fitting.py:240-258) and find_image_background."""
    R = ref_loader.load_reference()
    load, crop = ref_loader.load_io()
    d = {}
    for name in NORM_CASES:
        im = case_image(CASES[name])
        plain = quiet(R.fitting.fit_fov_image, im, "647", th_seed=600, verbose=False)
        loc = quiet(R.fitting.fit_fov_image, im, "647", th_seed=600, normalize_local=True, verbose=False)
        glo = quiet(R.fitting.fit_fov_image, im, "647", th_seed=600, normalize_background=True, verbose=False)
        d[name + "_plain"] = plain
        d[name + "_local"] = loc
        d[name + "_global"] = glo
        d[name + "_back"] = np.float64(quiet(load.find_image_background, im))
        d[name + "_back_b25_i3"] = np.float64(quiet(load.find_image_background, im, bin_size=25, max_iter=3))
        backs = []
        for pt in plain:
            c = crop.generate_neighboring_crop(pt[1:4], crop_size=10, single_im_size=np.array(im.shape))
            backs.append(load.find_image_background(im[c.to_slices()]))
        d[name + "_backs"] = np.array(backs, dtype=np.float64)
    sp = special_background_images()
    for k, im in sp.items():
        d["special_" + k] = np.float64(quiet(load.find_image_background, im))
        d["special_i1_" + k] = np.float64(quiet(load.find_image_background, im, max_iter=1))
    np.savez_compressed(os.path.join(OUT, "norm.npz"), **d)
    meta["norm_cases"] = NORM_CASES


def chain_golden(meta):
    """io_tools/load.py:166-522 correct_fov_image run by the reference on a synthetic .dax movie."""
    import tempfile
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
    import conftest as T
    ref_loader.load_corrections()
    load, _ = ref_loader.load_io()
    case = T.build_chain_case()
    d = {"raw_crc": crc(case["raw"])}
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "movie.dax")
        T.write_dax(path, case["raw"])
        for name in T.CHAIN_VARIANTS:
            sel, kw = T.chain_kwargs(case, name)
            out = quiet(load.correct_fov_image, path, sel, **kw)
            for ch, im in zip(sel, out[0]):
                d["%s_%s" % (name, ch)] = im
    np.savez_compressed(os.path.join(OUT, "chain.npz"), **d)
    meta["chain_variants"] = list(T.CHAIN_VARIANTS)


def chromfn_inputs():
    rng = np.random.RandomState(12)
    orders = np.array([1, 2, 2])
    ncol = {0: 1, 1: 4, 2: 10}
    consts = [rng.randn(ncol[int(o)]) * (10.0 ** (-2 * np.arange(ncol[int(o)]) / ncol[int(o)] - 1)) for o in orders]
    info = {'constants': consts, 'fitting_orders': orders, 'ref_center': np.array([15., 1024., 1024.])}
    coords = rng.rand(40, 3) * np.array([30., 2048., 2048.])
    spots = rng.rand(25, 11).astype(np.float32) * 100
    drift = np.array([0.4, -1.7, 2.2], dtype=np.float32)
    return info, coords, spots, drift


def chromfn_golden(meta):
    """correction_tools/chromatic.py:41-143 generate_chromatic_function on fixed inputs."""
    ch = ref_loader.load_chromatic()
    info, coords, spots, drift = chromfn_inputs()
    d = {}
    f = ch.generate_chromatic_function(info, drift)
    d["coords"], d["spots"] = f(coords), f(spots)
    f0 = ch.generate_chromatic_function(info, None)
    d["coords_nodrift"] = f0(coords)
    fd = ch.generate_chromatic_function(None, drift)
    d["drift_only"] = fd(spots)
    d["poly2"] = ch.generate_polynomial_data(coords[:7], 2)
    np.savez_compressed(os.path.join(OUT, "chromfn.npz"), **d)


def daxp_golden(meta):
    """classes/preprocess.py:337-1260 DaxProcesser, step by step, run by the reference on the synthetic movie of the
    chain case; the images after every step and the fitted spots are the fixtures."""
    import tempfile
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
    import conftest as T
    ref_loader.load_corrections()
    ns = ref_loader.load_reference()
    pre = sys.modules["IA3.classes.preprocess"]
    case = T.build_chain_case()
    chs = case["chs"]
    size = [case["Z"], case["X"], case["Y"]]
    d = {}

    def put(key, im):   # CRC of the whole stack + 4096 sampled voxels (enough to localise a mismatch)
        d[key + "_crc"] = crc(im)
        d[key + "_smp"] = samples(im)

    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "movie.dax")
        T.write_dax(path, case["raw"])
        for tag, kw in (("a", dict(rescale=True, illum64=False)), ("b", dict(rescale=False, illum64=True))):
            p = pre.DaxProcesser(path, Channels=chs, DriftChannel='488', verbose=False)
            quiet(p._load_image, ImSize=size, NbufferFrame=case["nb"])
            quiet(p._corr_hot_pixels_3D)
            for c in chs:
                put("%s_hot_%s" % (tag, c), getattr(p, "im_" + c))
            quiet(p._corr_bleedthrough, correction_pf=case["bleed"], rescale=kw["rescale"])
            for c in chs[:3]:
                put("%s_bleed_%s" % (tag, c), getattr(p, "im_" + c))
            illum = {k: (a.astype(np.float64) if kw["illum64"] else a) for k, a in case["illum"].items()}
            quiet(p._corr_illumination, correction_pf=illum, rescale=kw["rescale"])
            for c in chs:
                put("%s_illum_%s" % (tag, c), getattr(p, "im_" + c))
            quiet(p._warp_image, drift=np.array(case["drift"]), chromatic_pf=case["chrom"])
            for c in chs:
                put("%s_warp_%s" % (tag, c), getattr(p, "im_" + c))
            if tag == "a":
                quiet(p._gaussian_highpass, correction_channels=['750'])
                put("a_highpass_750", p.im_750)
                quiet(p._fit_spots, fit_channels=['647', '561'], th_seed=300)
                for c in ('647', '561'):
                    d["a_spots_%s" % c] = np.array(getattr(p, "spots_" + c))
    np.savez_compressed(os.path.join(OUT, "daxp.npz"), **d)


def profiles_golden(meta):
    """io_tools/load.py:553-640 load_correction_profile: the file names the reference opens (np.load / pickle.load are
    replaced by recorders) and the shapes / keys it returns."""
    import pickle
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
    from conftest import profile_name_cases
    load, _ = ref_loader.load_io()
    opened = []

    def fake_np_load(path, allow_pickle=False):
        opened.append(os.path.basename(path))
        return np.zeros(3 * 3 * 64 * 96, np.float32)

    real_np_load, real_open, real_pk = np.load, load.open if hasattr(load, "open") else None, pickle.load
    d = {}
    try:
        load.np.load = fake_np_load
        load.pickle.load = lambda f: {"constants": "const"}
        load.open = lambda path, mode="rb": opened.append(os.path.basename(path)) or None
        for i, (typ, kw) in enumerate(profile_name_cases()):
            del opened[:]
            pf = load.load_correction_profile(typ, correction_folder="/corr", **kw)
            d["names_%d" % i] = np.array(opened)
            d["keys_%d" % i] = np.array(sorted(pf.keys())) if isinstance(pf, dict) else np.array(pf.shape)
            d["none_%d" % i] = np.array([k for k in sorted(pf.keys()) if pf[k] is None]) if isinstance(pf, dict) else np.array([])
    finally:
        load.np.load = real_np_load
        load.pickle.load = real_pk
        if real_open is None:
            del load.open
    np.savez_compressed(os.path.join(OUT, "profile_names.npz"), **d)
    print({k: v.tolist() for k, v in d.items()})


def seg_golden(meta):
    """classes/preprocess.py:1093-1153 DaxProcesser._fit_spots_by_segmentation run by the reference on the movie of
    the chain case with the label image of tests/conftest.py::seg_labels."""
    import tempfile
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
    import conftest as T
    ref_loader.load_corrections()
    ref_loader.load_partition()
    pre = sys.modules["IA3.classes.preprocess"]
    case = T.build_chain_case()
    size = [case["Z"], case["X"], case["Y"]]
    lab = T.seg_labels(size)
    d = {"lab_crc": crc(lab)}
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "movie.dax")
        T.write_dax(path, case["raw"])
        p = pre.DaxProcesser(path, Channels=case["chs"], DriftChannel='488', verbose=False)
        quiet(p._load_image, ImSize=size, NbufferFrame=case["nb"])
        quiet(p._corr_hot_pixels_3D)
        with open(os.devnull, "w") as nul, contextlib.redirect_stderr(nul):   # tqdm bar
            quiet(p._fit_spots_by_segmentation, '647', lab, th_seed=300, segment_search_radius=3)
            d["spots_647"], d["ids_647"] = np.array(p.spots_647), np.array(p.spots_cell_ids_647)
            p.drift = np.array(case["drift"])
            out = quiet(p._fit_spots_by_segmentation, '750', lab, th_seed=300, num_spots=2, save_attrs=False)
            d["spots_750"], d["ids_750"] = np.array(out[0]), np.array(out[1])
            out = quiet(p._fit_spots_by_segmentation, '561', (lab == 4) * 4, th_seed=300, save_attrs=False)
            d["spots_561"], d["ids_561"] = np.array(out[0]), np.array(out[1])
    np.savez_compressed(os.path.join(OUT, "seg.npz"), **d)
    print("seg", {k: np.shape(v) for k, v in d.items()})


def fit_case_golden(R, name, spec):
    """Seed tables, first / final rows, fit_fov_image tables and a few Voronoi voxel sets of one case, all from the
    reference's own functions."""
    fit, F4 = R.fitting, R.F4
    im = case_image(spec)
    d = {}
    d["seeds_h"] = fit.get_seeds(im, th_seed=600, return_h=True)
    d["seeds_nodyn"] = fit.get_seeds(im, th_seed=600, use_dynamic_th=False, return_h=True)
    d["seeds_hi_th"] = fit.get_seeds(im, th_seed=9000, return_h=True, min_dynamic_seeds=5)
    d["seeds_nohot"] = fit.get_seeds(im, th_seed=600, remove_hot_pixel=False, return_h=True)
    d["seeds_top10"] = fit.get_seeds(im, th_seed=600, max_num_seeds=10, return_h=True)
    cen = [s // 2 for s in spec["shape"]]
    d["sel_center"] = np.array(cen)
    d["seeds_sel"] = fit.get_seeds(im, th_seed=600, sel_center=cen, seed_radius=20, return_h=True)
    d["seeds_edge0"] = fit.get_seeds(im, th_seed=600, min_edge_distance=0, return_h=True)
    # fitting
    seeds = fit.get_seeds(im, th_seed=600)
    fitter = F4.iter_fit_seed_points(im, seeds.T, radius_fit=5)
    quiet(fitter.firstfit)
    d["first_ps"] = np.array(fitter.ps, dtype=np.float32)
    d["first_nvox"] = np.array([len(g[0]) for g in fitter.gparms])
    quiet(fitter.repeatfit)
    d["final_ps"] = np.array(fitter.ps, dtype=np.float32)
    d["n_iter"] = np.array(fitter.n_iter)
    d["table"] = quiet(fit.fit_fov_image, im, "647", th_seed=600, max_num_seeds=None, verbose=False)
    d["table_max20"] = quiet(fit.fit_fov_image, im, "647", th_seed=600, max_num_seeds=20, verbose=False)
    d["centers"] = quiet(fit.get_centers, im, th_seed=600)
    d["sparse"] = fit.select_sparse_centers(d["centers"], distance_th=25)
    # a few voxel sets (Voronoi cells) of the first fit
    for k in range(min(3, len(fitter.gparms))):
        d["gp%d_X" % k] = np.array(fitter.gparms[k][1])
        d["gp%d_im" % k] = np.array(fitter.gparms[k][0])
    np.savez_compressed(os.path.join(OUT, "fit_%s.npz" % name), **d)
    print(name, "seeds", len(d["seeds_h"]), "table", d["table"].shape, "n_iter", fitter.n_iter)


SEEDOPT_CASES = ["c1_f32", "c1_u16", "clu_f32"]


def seed_mask_for(shape):
    """The mask of the seed_mask fixtures (mirrored by tests): a slanted half-space plus a box, as float32 0/1."""
    z, x, y = np.meshgrid(*[np.arange(n) for n in shape], indexing="ij")
    m = ((x + 2 * y) % 97 < 60) | ((z > shape[0] // 2) & (x < shape[1] // 3))
    return m.astype(np.float32)


def seedopts_golden(meta):
    """Options of get_seeds / fit_fov_image that the other fixtures leave at their defaults: the percentile threshold
    (spot_tools/fitting.py:75-76, scipy.stats.scoreatpercentile on the uncropped image) and seed_mask (:210-218)."""
    R = ref_loader.load_reference()
    fit = R.fitting
    d = {}
    for name in SEEDOPT_CASES:
        im = case_image(CASES[name])
        for per in (95, 99.5, 98):
            tag = "%s_per%s" % (name, str(per).replace(".", "p"))
            d[tag] = fit.get_seeds(im, use_percentile=True, th_seed_per=per, return_h=True)
            d[tag + "_nodyn"] = fit.get_seeds(im, use_percentile=True, th_seed_per=per, use_dynamic_th=False, return_h=True)
        d[name + "_per_sel"] = fit.get_seeds(im, use_percentile=True, th_seed_per=99.5, return_h=True,
                                             sel_center=[s // 2 for s in im.shape], seed_radius=25)
        d[name + "_per_table"] = quiet(fit.fit_fov_image, im, "647", use_percentile=True, th_seed_per=99.5,
                                       max_num_seeds=None, verbose=False)
        mask = seed_mask_for(im.shape)
        d[name + "_mask_table"] = quiet(fit.fit_fov_image, im, "647", th_seed=600, max_num_seeds=None, seed_mask=mask,
                                        verbose=False)
        d[name + "_mask_given"] = quiet(fit.fit_fov_image, im, "647", seeds=fit.get_seeds(im, th_seed=600, return_h=True),
                                        seed_mask=mask > 0, verbose=False)
    np.savez_compressed(os.path.join(OUT, "seedopts.npz"), **d)
    print("seedopts:", {k: v.shape for k, v in d.items() if k.endswith("table")})


def one_case(name):
    """python oracle/make_golden.py case:<name> — one fitting case, meta.json updated in place."""
    R = ref_loader.load_reference()
    fit_case_golden(R, name, CASES[name])
    mp = os.path.join(OUT, "meta.json")
    with open(mp) as f:
        meta = json.load(f)
    meta["cases"][name] = CASES[name]
    with open(mp, "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


def main():
    os.makedirs(OUT, exist_ok=True)
    R = ref_loader.load_reference()
    fit, F4 = R.fitting, R.F4
    meta = {"cases": CASES, "numpy": np.__version__}
    import scipy
    meta["scipy"] = scipy.__version__

    # ---------------- seeding + fitting tables --------------------------------------------
    for name, spec in CASES.items():
        fit_case_golden(R, name, spec)

    # ---------------- single-spot known answer (GaussianFit) --------------------------------
    shape = (30, 64, 64)
    c0 = np.array([[14.3, 30.6, 33.2]])
    im64 = np.full(shape, 100.0)
    synth.add_spots(im64, c0, np.array([2000.0]))
    im1 = im64.astype(np.float32)
    f1 = F4.iter_fit_seed_points(im1, np.array([[14.0], [31.0], [33.0]]), radius_fit=5)
    quiet(f1.firstfit)
    quiet(f1.repeatfit)
    np.savez_compressed(os.path.join(OUT, "single_spot.npz"), ps=np.array(f1.ps, dtype=np.float32),
                        n_iter=np.array(f1.n_iter), center=c0, shape=np.array(shape))

    # ---------------- filters: high-pass, hot pixels ----------------------------------------
    d = {}
    for name in ("c1_f32", "c1_u16", "hot_u16"):
        im = case_image(CASES[name])
        for sg, tr in ((3, 2), (5, 2)):
            hp = R.filter.gaussian_high_pass_filter(im, sg, tr)
            idx, val = samples(hp)
            d["hp_%s_s%d_crc" % (name, sg)] = crc(hp)
            d["hp_%s_s%d_idx" % (name, sg)] = idx
            d["hp_%s_s%d_val" % (name, sg)] = val
            d["hp_%s_s%d_sum" % (name, sg)] = np.float64(hp.astype(np.float64).sum())
        rh = R.filter.Remove_Hot_Pixels(im, dtype=im.dtype)
        d["rhp_%s_crc" % name] = crc(rh)
        d["rhp_%s_ndiff" % name] = np.int64((rh != im).sum())
        idx, val = samples(rh)
        d["rhp_%s_idx" % name] = idx
        d["rhp_%s_val" % name] = val
    np.savez_compressed(os.path.join(OUT, "filters.npz"), **d)

    # ---------------- warp --------------------------------------------------------------------
    d = {}
    drift = np.array([0.37, -2.6, 4.25])
    for name in ("c1_f32", "c1_u16"):
        im = case_image(CASES[name])[:, :96, :80]
        Z, X, Y = im.shape
        zz, xx, yy = np.meshgrid(np.arange(Z), np.arange(X), np.arange(Y), indexing="ij")
        field = np.stack([0.002 * (xx - X / 2), 0.01 * (yy - Y / 2) + 0.2, -0.008 * (xx - X / 2) + 0.005 * zz])
        for order, mode in ((1, "constant"), (3, "nearest"), (1, "nearest")):
            for use_field in (False, True):
                w = R.translate.warp_3d_image(im, drift, chromatic_profile=field if use_field else None,
                                              warp_order=order, border_mode=mode)
                key = "warp_%s_o%d_%s_f%d" % (name, order, mode, int(use_field))
                idx, val = samples(w)
                d[key + "_crc"] = crc(w)
                d[key + "_idx"] = idx
                d[key + "_val"] = val
    d["drift"] = drift
    np.savez_compressed(os.path.join(OUT, "warp.npz"), **d)

    # ---------------- drift: crops, fft3d_from2d, bead-path align_image, pairing --------------
    d = {}
    for k, size in enumerate(([30, 2048, 2048], [50, 2048, 2048], [30, 256, 256], [12, 100, 60])):
        d["crops_%d_size" % k] = np.array(size)
        d["crops_%d" % k] = R.alignment.generate_drift_crops(size)
    bshape = (30, 256, 256)
    true_d = np.array([1.3, -4.6, 7.25])
    ref, src, bc, bh = synth.make_bead_pair(bshape, 120, 21, true_d)
    d["bead_shape"] = np.array(bshape)
    d["bead_true_d"] = true_d
    d["fft3d"] = R.alignment_tools.fft3d_from2d(src, ref, gb=0, max_disp=128)
    d["fft3d_F4style_xy"] = np.array(R.alignment_tools.fftalign_2d(np.max(src, 0), np.max(ref, 0), max_disp=50))
    drift, flag = quiet(R.alignment.align_image, src, ref, use_autocorr=False,
                        correction_args={"single_im_size": list(bshape)}, verbose=False)
    d["align_beads_drift"] = np.array(drift)
    d["align_beads_flag"] = np.array(flag)
    # pairing on fitted centres of one crop
    crop = R.alignment.generate_drift_crops(list(bshape))[0]
    s = tuple(slice(*c) for c in crop)
    fa = dict(th_seed=300, use_dynamic_th=True, min_dynamic_seeds=10, max_num_seeds=200)
    ss = quiet(fit.fit_fov_image, src[s], "488", verbose=False, **fa)
    rs = quiet(fit.fit_fov_image, ref[s], "488", verbose=False, **fa)
    sc = fit.select_sparse_centers(ss[:, 1:4], 2.)
    rc = fit.select_sparse_centers(rs[:, 1:4], 2.)
    d["pair_src_cts"], d["pair_ref_cts"] = sc, rc
    rough = R.alignment_tools.fft3d_from2d(src[s], ref[s], gb=0, max_disp=np.max(src[s].shape) / 2)
    d["pair_rough"] = rough
    dr, pt, pr = R.matching.find_paired_centers(sc, rc, rough, cutoff=2.)
    d["pair_drift"], d["pair_tar"], d["pair_ref"] = dr, pt, pr
    if len(pr) > 3:
        dr2, pt2, pr2 = R.matching.check_paired_centers(pt, pr, outlier_sigma=1.5)
        d["check_drift"], d["check_tar"], d["check_ref"] = dr2, pt2, pr2
    np.savez_compressed(os.path.join(OUT, "drift.npz"), **d)

    legacy_golden(meta)
    norm_golden(meta)
    chain_golden(meta)
    chromfn_golden(meta)
    daxp_golden(meta)
    seg_golden(meta)
    profiles_golden(meta)
    seedopts_golden(meta)

    with open(os.path.join(OUT, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    if len(sys.argv) > 1:   # regenerate single fixtures: python oracle/make_golden.py seg_golden ...
        for _name in sys.argv[1:]:
            if _name.startswith("case:"):
                one_case(_name[5:])
            else:
                globals()[_name]({})
    else:
        main()
