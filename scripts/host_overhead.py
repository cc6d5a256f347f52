"""ms per FOV from one stream with the HIP-event profile on / off and 1 / 4 distinct resident FOVs (developer probe)."""
import sys, os, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis3_amd import synth, _lib as L
lib = L.lib(); L.check(lib.ia3_init(0))
ims = [synth.make_fov((50, 2048, 2048), 5000, 3 + k)[0] for k in range(4)]
sts = [L.DeviceStack.upload(im) for im in ims]
sp, keep = L.make_seed_params(600.0, max_num_seeds=None); fp = L.make_fit_params()
rows = np.empty((16384, 11), np.float32); nr, ns, ni = C.c_int(0), C.c_int(0), C.c_int(0)
def run(n, pool):
    for i in range(n):
        L.check(lib.ia3_fit_fov_dev(sts[i % pool]._h, C.byref(sp), C.byref(fp), L.ptr(rows), len(rows), C.byref(nr), C.byref(ns), C.byref(ni)))
for rep in range(2):
    for prof in (1, 0):
        for pool in (1, 4):
            L.profile_enable(bool(prof)); L.profile_collect()
            run(4, pool); lib.ia3_sync(); t0 = time.perf_counter()
            run(20, pool); lib.ia3_sync(); d = (time.perf_counter() - t0) / 20
            L.profile_collect(); L.profile_enable(False)
            print("profile=%d pool=%d: %.3f ms/FOV" % (prof, pool, d * 1e3), flush=True)
