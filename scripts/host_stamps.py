"""Host time stamps (IA3_DEBUG_TIMES) of a few per-FOV calls on one resident float32 bench FOV: where the host is while
the device idles between and inside steps (developer probe).  Run with IA3_DEBUG_TIMES=1; stamps go to stderr."""
import sys, os, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis3_amd import synth, _lib as L
lib = L.lib(); L.check(lib.ia3_init(0))
im = synth.make_fov((50, 2048, 2048), 5000, 3)[0]
st = L.DeviceStack.upload(im)
sp, keep = L.make_seed_params(600.0, max_num_seeds=None); fp = L.make_fit_params()
rows = np.empty((16384, 11), np.float32); nr, ns, ni = C.c_int(0), C.c_int(0), C.c_int(0)
for i in range(12):
    t0 = time.perf_counter()
    L.check(lib.ia3_fit_fov_dev(st._h, C.byref(sp), C.byref(fp), L.ptr(rows), len(rows), C.byref(nr), C.byref(ns), C.byref(ni)))
    sys.stderr.write("PY step %d took %.1f us\n" % (i, (time.perf_counter() - t0) * 1e6))
