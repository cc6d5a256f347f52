import os
import sys
import json
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _meta():
    with open(os.path.join(GOLDEN, "meta.json")) as f:
        return json.load(f)


def build_case(name):
    """Regenerate the input stack of a golden case (mirror of oracle/make_golden.py::case_image)."""
    from imageanalysis3_amd import synth
    spec = _meta()["cases"][name]
    im, c, h = synth.make_fov(tuple(spec["shape"]), spec["n"], spec["seed"], layout=spec["layout"],
                              dtype=np.dtype(spec["dtype"]),
                              **{k: spec[k] for k in ("n_territories", "min_sep", "margin") if k in spec})
    for (x, y, v) in spec.get("hot_columns", []):
        im[:, x, y] = v
    return im


def seed_mask_for(shape):
    """The mask of the seed_mask fixtures (mirror of oracle/make_golden.py::seed_mask_for)."""
    z, x, y = np.meshgrid(*[np.arange(n) for n in shape], indexing="ij")
    m = ((x + 2 * y) % 97 < 60) | ((z > shape[0] // 2) & (x < shape[1] // 3))
    return m.astype(np.float32)


def load_golden(fname):
    return dict(np.load(os.path.join(GOLDEN, fname), allow_pickle=False))


def golden_samples_idx(size, n=4096, seed=99):
    from imageanalysis3_amd import synth
    return (synth.uniform01(seed, 11, np.arange(n)) * size).astype(np.int64)


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def cases():
    return _meta()["cases"]


def build_legacy():
    """Input of tests/golden/legacy.npz (mirror of oracle/make_golden.py::legacy_golden)."""
    from imageanalysis3_amd import synth
    m = _meta()["legacy"]
    spec = m["image"]
    im, c, h = synth.make_fov(tuple(spec["shape"]), spec["n"], spec["seed"], layout=spec["layout"],
                              dtype=np.dtype(spec["dtype"]), margin=tuple(spec["margin"]),
                              n_territories=spec["n_territories"])
    return im, m


def special_background_images():
    """Mirror of oracle/make_golden.py::special_background_images."""
    rng = np.random.RandomState(4)
    d = {}
    d["const"] = np.full((6, 12, 12), 500, np.uint16)
    d["const0"] = np.zeros((6, 12, 12), np.uint16)
    d["plateau"] = np.repeat(np.array([100, 110, 120, 130], np.uint16), 216).reshape(6, 12, 12)
    d["two_peaks_tie"] = np.repeat(np.array([100, 300, 500, 700], np.uint16), [300, 132, 300, 132]).reshape(6, 12, 12)
    d["top_edge"] = np.full((6, 12, 12), 65530, np.uint16)
    d["above_range"] = np.full((6, 12, 12), 65534, np.uint16)
    x = rng.normal(420., 35., size=(8, 20, 20))
    d["noise_u16"] = np.clip(x, 0, 65535).astype(np.uint16)
    d["noise_f32"] = x.astype(np.float32)
    y = x.astype(np.float32).copy(); y[0, :3, :3] = np.nan
    d["nan_f32"] = y
    d["sparse"] = (rng.randint(0, 6000, size=(4, 8, 8)) * 10).astype(np.uint16)
    return d


def build_chain_case():
    """Synthetic 4-colour .dax movie + correction profiles for the correct_fov_image chain
    (mirror of oracle/make_golden.py::chain_golden).  Returns a dict."""
    from imageanalysis3_amd import synth
    Z, X, Y = 12, 64, 64
    chs = ['750', '647', '561', '488']
    nb = 2
    ims = [synth.make_fov((Z, X, Y), 6, 30 + i, dtype=np.uint16, margin=(2, 6, 6))[0] for i in range(4)]
    # hot columns: an isolated one, two adjacent ones (later candidates see earlier replacements), one on the border
    for (x, y, v) in ((20, 21, 9000), (40, 30, 12000), (40, 31, 11000), (0, 5, 8000)):
        for im in ims[:3]:
            im[:, x, y] = v
    frames = nb + Z * 4 + nb
    raw = np.zeros((frames, X, Y), np.uint16)
    for i in range(4):
        start = nb + (i - nb) % 4
        raw[start:start + Z * 4:4] = ims[i]
    rng = np.random.RandomState(0)
    illum = {c: (0.6 + 0.4 * rng.rand(X, Y)).astype(np.float32) for c in chs}
    bleed = (np.eye(3)[:, :, None, None] + 0.05 * rng.rand(3, 3, X, Y)).astype(np.float32)
    chrom = {c: (0.3 * rng.randn(3, Z, X, Y)).astype(np.float32) for c in chs[:3]}
    chrom['647'] = None
    return dict(Z=Z, X=X, Y=Y, chs=chs, nb=nb, raw=raw, illum=illum, bleed=bleed, chrom=chrom,
                drift=[0.3, -1.2, 2.5])


def h5_helper_inputs():
    """Images / spot tables of the save-file helper fixtures (oracle/make_golden_h5.py::helpers_golden)."""
    rng = np.random.RandomState(5)
    ims = [rng.randint(0, 60000, size=(4, 8, 8)).astype(np.uint16) for _ in range(3)]
    spots = [rng.rand(n, 11).astype(np.float32) * 100 for n in (3, 7, 2)]
    raw = [s + np.float32(0.5) for s in spots]
    return ims, spots, raw


def batch_inputs():
    """Movie + correction / fitting arguments of the batch_process_image_to_spots fixtures
    (oracle/make_golden_h5.py::batch_golden): the chain case, dense chromatic fields for the warped variant and
    polynomial constants for the unwarped one."""
    case = build_chain_case()
    size = [case["Z"], case["X"], case["Y"]]
    corr = dict(single_im_size=size, all_channels=case["chs"], num_buffer_frames=case["nb"], num_empty_frames=0,
                corr_channels=case["chs"][:3], illumination_profile=case["illum"], bleed_profile=case["bleed"],
                chromatic_profile=case["chrom"])
    rng = np.random.RandomState(21)
    consts = {}
    for c in case["chs"][:3]:
        consts[c] = None if c == '647' else {'constants': [rng.randn(4) * np.array([0.3, 1e-3, 1e-4, 1e-4]) for _ in range(3)],
                                             'fitting_orders': np.array([1, 1, 1]),
                                             'ref_center': np.array([size[0] / 2., size[1] / 2., size[2] / 2.])}
    corr_nowarp = dict(corr, chromatic_profile=consts)
    fit = dict(max_num_seeds=20, seeding_kwargs={})
    return case, size, corr, corr_nowarp, fit


def profile_name_cases():
    """(corr_type, kwargs) calls of the load_correction_profile fixtures (oracle/make_golden.py::profiles_golden)."""
    chs = ['750', '647', '561', '488', '405']
    return [("illumination", dict(corr_channels=['750', '561', '488'], all_channels=chs, im_size=[12, 64, 96])),
            ("bleedthrough", dict(corr_channels=['561', '750', '647'], all_channels=chs, im_size=[12, 64, 96])),
            ("chromatic", dict(corr_channels=['750', '647', '561'], all_channels=chs, ref_channel='647', im_size=[12, 64, 96])),
            ("chromatic_constants", dict(corr_channels=['750', '647'], all_channels=chs, ref_channel='647', im_size=[30, 2048, 2048])),
            ("Illumination", dict(corr_channels=['405'], all_channels=chs, im_size=[50, 2048, 2048]))]


def seg_labels(shape):
    """Label image for the segmentation-driven fit fixtures: a quadrant box (1), a central ellipsoid (2), a thin
    slab along one edge (3) and a 3-voxel speck (4)."""
    Z, X, Y = shape
    lab = np.zeros(shape, np.int32)
    lab[1:Z - 1, 2:X // 2 - 2, 3:Y // 2 - 1] = 1
    z, x, y = np.meshgrid(np.arange(Z), np.arange(X), np.arange(Y), indexing='ij')
    ell = ((z - Z / 2.) / (Z / 2.2)) ** 2 + ((x - 0.66 * X) / (X / 4.)) ** 2 + ((y - 0.6 * Y) / (Y / 3.5)) ** 2 <= 1
    lab[ell & (lab == 0)] = 2
    lab[:, X - 4:X, 0:Y // 2] = 3
    lab[Z // 2, 5, Y - 6:Y - 3] = 4
    return lab


def write_dax(path, raw):
    raw.astype('<u2').tofile(path)
    with open(path[:-4] + ".inf", "w") as f:
        f.write("information file for\n%s\nframe dimensions = %d x %d\nnumber of frames = %d\n little endian\n"
                % (path, raw.shape[2], raw.shape[1], raw.shape[0]))


CHAIN_VARIANTS = {
    "full": dict(z_shift_corr=True, verbose=True),
    "silent_no_warp": dict(z_shift_corr=False, verbose=False),
    "highpass": dict(gaussian_highpass=True, verbose=True, bleed_corr=False, chromatic_corr=False),
    "no_drift_647_only": dict(verbose=True, drift=None, sel=['647']),
    "no_hot_f64_illum": dict(verbose=True, hot_pixel_corr=False, illum64=True),
}


def chain_kwargs(case, name):
    v = dict(CHAIN_VARIANTS[name])
    sel = v.pop("sel", ['750', '647'])
    illum = case["illum"]
    if v.pop("illum64", False):
        illum = {k: a.astype(np.float64) for k, a in illum.items()}
    kw = dict(single_im_size=[case["Z"], case["X"], case["Y"]], all_channels=case["chs"],
              num_buffer_frames=case["nb"], num_empty_frames=0, drift=case["drift"], corr_channels=case["chs"][:3],
              illumination_profile=illum, bleed_profile=case["bleed"], chromatic_profile=case["chrom"])
    kw.update(v)
    return sel, kw
