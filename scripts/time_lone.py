"""Lone-image latency (one ia3_fit_fov_dev per image, nothing else on the device) of the bench's uint16 FOV and of the
crowded layout-B field, with the unchanged-neighbour shortcut of the refit sweeps on and off (developer tool)."""
import ctypes as C, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis3_amd import synth, _lib as L
lib = L.lib(); L.check(lib.ia3_init(0))
sp, keep = L.make_seed_params(600.0, max_num_seeds=None); fp = L.make_fit_params()
rows = np.empty((65536, 11), np.float32); nr, ns, ni = C.c_int(0), C.c_int(0), C.c_int(0)
cases = [("uint16 isolated (bench u16_0, seed 40)", dict(seed=40, dtype=np.uint16)),
         ("float32 clustered (seed 50)", dict(seed=50, dtype=np.float32, layout="clustered")),
         ("uint16 clustered (seed 50)", dict(seed=50, dtype=np.uint16, layout="clustered"))]
for name, kw in cases:
    seed = kw.pop("seed")
    im = synth.make_fov((50, 2048, 2048), 5000, seed, **kw)[0]
    with L.DeviceStack.upload(im) as st:
        tabs = {}
        for memo in (1, 0, 1):
            L.check(lib.ia3_set_tuning(14, memo))
            ts = []
            for rep in range(3):
                lib.ia3_sync(); t0 = time.perf_counter()
                L.check(lib.ia3_fit_fov_dev(st._h, C.byref(sp), C.byref(fp), L.ptr(rows), len(rows), C.byref(nr), C.byref(ns), C.byref(ni)))
                ts.append(time.perf_counter() - t0)
            a = [C.c_int64(0), C.c_int64(0), C.c_int64(0)]
            lib.ia3_fit_fov_stats(C.byref(a[0]), C.byref(a[1]), C.byref(a[2]))
            tabs[memo] = rows[:nr.value].copy()
            print("%s  memo %d: %d seeds %d rows %d sweeps  fits %d nfev %d  min %.2f ms" % (
                name, memo, ns.value, nr.value, ni.value, a[0].value, a[1].value, min(ts) * 1e3), flush=True)
        print("   tables identical:", np.array_equal(tabs[0], tabs[1]), flush=True)
L.check(lib.ia3_set_tuning(14, 1))
