"""The reference's calling pattern: the operators run inside forked ``mp.Pool(...).starmap(chunksize=1)`` children
(classes/field_of_view.py:1129-1142, worker classes/batch_functions.py:60).  libia3 creates its HIP state lazily per
process (runtime.cpp do_init: a pid change starts over), so a parent that has not touched the GPU can fork workers that
each open the device themselves.  The scenario runs in a fresh interpreter (a process that has initialised HIP must
neither fork workers that use the GPU nor be replaced by exec)."""
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, json
import multiprocessing as mp
import numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))

def fit_task(name, th):
    from conftest import build_case
    from imageanalysis3_amd.spot_tools.fitting import fit_fov_image
    t = fit_fov_image(build_case(name), "647", th_seed=th, max_num_seeds=None, verbose=False)
    return name, os.getpid(), np.asarray(t)

def chain_task(variant, tmp):
    from conftest import build_chain_case, chain_kwargs, write_dax
    from imageanalysis3_amd.io_tools.load import correct_fov_image
    case = build_chain_case()
    sel, kw = chain_kwargs(case, variant)
    path = os.path.join(tmp, "movie_%%d.dax" %% os.getpid())
    write_dax(path, case["raw"])
    out = correct_fov_image(path, sel, **kw)
    return variant, os.getpid(), [np.asarray(a) for a in out[0]], sel

if __name__ == "__main__":
    from conftest import load_golden
    import imageanalysis3_amd                      # package import only: no HIP call in the parent before the fork
    tmp = sys.argv[1]
    with mp.get_context("fork").Pool(2) as pool:
        fits = pool.starmap(fit_task, [("c1_f32", 600), ("c1_u16", 600), ("m_f32", 600), ("hot_u16", 600)], chunksize=1)
        chains = pool.starmap(chain_task, [("full", tmp), ("highpass", tmp)], chunksize=1)
    pids = set(p for _, p, _ in fits) | set(p for _, p, _, _ in chains)
    assert os.getpid() not in pids and len(pids) == 2, pids
    worst = 0.0
    for name, pid, t in fits:
        ref = load_golden("fit_%%s.npz" %% name)["table"]
        assert t.shape == ref.shape, (name, t.shape, ref.shape)
        from scipy.spatial import cKDTree
        d, j = cKDTree(ref[:, 1:4]).query(t[:, 1:4])
        assert d.max() < 0.05 and len(np.unique(j)) == len(j)
        rel = np.abs(t[:, :8].astype(float) - ref[j, :8]) / np.abs(ref[j, :8])
        worst = max(worst, float(rel.max()))
    assert worst <= 1e-4, worst
    g = load_golden("chain.npz")
    for variant, pid, ims, sel in chains:
        for ch, im in zip(sel, ims):
            assert np.array_equal(im, g["%%s_%%s" %% (variant, ch)]), (variant, ch)
    # the parent itself can still open the device afterwards (its own lazy init)
    n, p, t = fit_task("c1_f32", 600)
    assert p == os.getpid() and t.shape == load_golden("fit_c1_f32.npz")["table"].shape
    print(json.dumps({"ok": True, "workers": sorted(pids), "worst_rel": worst}))
'''


def test_operators_inside_forked_pool_workers(tmp_path):
    script = tmp_path / "forked_pool.py"
    script.write_text(CHILD % {"root": ROOT})
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, str(script), str(tmp_path)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert '"ok": true' in r.stdout
