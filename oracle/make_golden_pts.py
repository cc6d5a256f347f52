"""TEST INFRASTRUCTURE ONLY — fixture of the image-free bead alignment (reference: alignment_tools.py:356-419
``translation_align_pts`` and correction_tools/alignment.py:139-216 ``align_beads(use_fft=False)``), made by running the
reference's own functions through ``oracle/ref_loader.py`` on the bead centres of ``tests/golden/drift.npz`` and on a
seeded random point set.  Writes ``tests/golden/alignpts.npz``.  Runs only where /root/reference exists."""
import contextlib, io, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader

R = ref_loader.load_reference()
G = os.path.join(HERE, "..", "tests", "golden")
drift = np.load(os.path.join(G, "drift.npz"))
d = {}
cases = {"beads": (drift["pair_ref_cts"], drift["pair_src_cts"])}
rng = np.random.RandomState(17)
fix = rng.uniform(0, [30, 200, 200], size=(40, 3))
keep = rng.rand(40) < 0.8
tar = fix[keep] + np.array([1.4, -6.3, 9.2]) + rng.normal(0, 0.05, size=(int(keep.sum()), 3))
tar = np.concatenate([tar, rng.uniform(0, [30, 200, 200], size=(6, 3))])   # a few points without a partner
cases["random"] = (fix, tar[rng.permutation(len(tar))])
for name, (a, b) in cases.items():
    d[name + "_fix"], d[name + "_tar"] = a, b
    with contextlib.redirect_stdout(io.StringIO()):
        t, pf, pt = R.alignment_tools.translation_align_pts(a, b, cutoff=2., return_pts=True)
        t3 = R.alignment_tools.translation_align_pts(a, b, cutoff=1., xyz_res=2)
        out = R.alignment.align_beads(b, a, use_fft=False, match_distance_th=2., check_paired_cts=True,
                                      outlier_sigma=1.5, return_paired_cts=True, verbose=False)
    d[name + "_t"], d[name + "_pf"], d[name + "_pt"], d[name + "_t_res2"] = t, pf, pt, t3
    d[name + "_ab_drift"], d[name + "_ab_tar"], d[name + "_ab_ref"] = out
np.savez_compressed(os.path.join(G, "alignpts.npz"), **d)
print({k: np.shape(v) for k, v in d.items()})
