"""Fused per-FOV pass on a uint16 2048x2048x50 stack (production dtype): timing + sanity (developer tool)."""
import ctypes as C, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis3_amd import synth, _lib as L
lib = L.lib(); L.check(lib.ia3_init(0))
im, c, h = synth.make_fov((50, 2048, 2048), 5000, 3, dtype=np.uint16)
st = L.DeviceStack.upload(im)
sp, keep = L.make_seed_params(600.0, max_num_seeds=None); fp = L.make_fit_params()
rows = np.empty((65536, 11), np.float32); nr, ns, ni = C.c_int(0), C.c_int(0), C.c_int(0)
for rep in range(4):
    lib.ia3_sync(); t0 = time.perf_counter()
    L.check(lib.ia3_fit_fov_dev(st._h, C.byref(sp), C.byref(fp), L.ptr(rows), len(rows), C.byref(nr), C.byref(ns), C.byref(ni)))
    lib.ia3_sync(); dt = time.perf_counter() - t0
    print("uint16: %d seeds, %d rows, %d sweeps, %.2f ms" % (ns.value, nr.value, ni.value, dt * 1e3))
r = rows[:nr.value]
from scipy.spatial import cKDTree
d, j = cKDTree(c).query(r[:, 1:4])
print("matched to injected centres: median |d| %.3f px, max %.3f px, found %d of %d" % (np.median(d), d.max(), len(np.unique(j)), len(c)))
L.profile_enable(True); L.profile_collect()
L.check(lib.ia3_fit_fov_dev(st._h, C.byref(sp), C.byref(fp), L.ptr(rows), len(rows), C.byref(nr), C.byref(ns), C.byref(ni)))
prof = L.profile_collect()
print({k: round(v[1], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])})
hh = C.c_void_p(); out = np.empty((65536, 4)); nn = C.c_int(0); th = C.c_double(0)
L.check(lib.ia3_dog_seed_dev(st._h, C.byref(sp), L.dptr(out), len(out), C.byref(nn), C.byref(th)))
seeds = np.ascontiguousarray(out[:nn.value, :3])
L.check(lib.ia3_fit_create(st._h, L.dptr(seeds), len(seeds), C.byref(fp), C.byref(hh)))
L.check(lib.ia3_fit_first(hh)); L.check(lib.ia3_fit_repeat(hh, None))
a, b = C.c_int64(0), C.c_int64(0); lib.ia3_fit_stats(hh, C.byref(a), C.byref(b)); print("fits", a.value, "nfev", b.value)
nf = np.empty(len(seeds), np.int32)
