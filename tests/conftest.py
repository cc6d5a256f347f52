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
