"""Host-side split of one bench step (developer tool): IA3_DEBUG_TIMES=1 python scripts/step_times.py"""
import ctypes as C, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis3_amd import synth, _lib as L
lib = L.lib(); L.check(lib.ia3_init(0))
im, c, h = synth.make_fov((50, 2048, 2048), 5000, 3)
st = L.DeviceStack.upload(im)
sp, keep = L.make_seed_params(600.0, max_num_seeds=None); fp = L.make_fit_params()
rows = np.zeros((16384, 11), np.float32); nr, ns, ni = C.c_int(0), C.c_int(0), C.c_int(0)
for rep in range(6):
    t0 = time.perf_counter()
    L.check(lib.ia3_fit_fov_dev(st._h, C.byref(sp), C.byref(fp), L.ptr(rows), len(rows), C.byref(nr), C.byref(ns), C.byref(ni)))
    print("step %.1f us" % (1e6 * (time.perf_counter() - t0)), file=sys.stderr)
