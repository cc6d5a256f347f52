import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import build_legacy, load_golden
from imageanalysis3_amd import visual_tools as vt
from imageanalysis3_amd.External import Fitting_v3
np.set_printoptions(linewidth=200, precision=5, suppress=True)
im, m = build_legacy()
g = load_golden("legacy.npz")
sa = tuple(m["seeding"]["default"][:-1]) + (False,)
fa = tuple(m["fitting_args"])
norm = np.nanmedian(im)
for i in (0, 1, 3):
    s = vt.get_seed_in_distance(im, g["coords"][i], *sa)
    print("seeds", s.tolist())
    f = Fitting_v3.iter_fit_seed_points(im, s.T, *fa)
    f.firstfit()
    a = np.array(f.ps); r = g["first_%d" % i]
    print("first rel", np.abs(a[:, :8] - r[:, :8]).max(0) / np.abs(r[:, :8]).max(0), "nvox", f.nvox)
    f.repeatfit()
    a = np.array(f.ps); r = g["fit_%d" % i].copy(); r[:, 0] *= norm
    print("final rel", (np.abs(a[:, :8] - r[:, :8]) / np.abs(r[:, :8])).max(1), "n_iter", f.n_iter, int(g["n_iter_%d" % i]))
