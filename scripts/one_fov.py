"""One fused FOV pass (developer tool for rocprofv3 counter runs)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis3_amd import synth, _lib as L
lib = L.lib(); L.check(lib.ia3_init(0))
im, c, h = synth.make_fov((50, 1024, 1024), 2500, 3)
st = L.DeviceStack.upload(im)
sp, keep = L.make_seed_params(600.0, max_num_seeds=None); fp = L.make_fit_params()
rows = np.empty((65536, 11), np.float32); nr, ns, ni = C.c_int(0), C.c_int(0), C.c_int(0)
for _ in range(2):
    L.check(lib.ia3_fit_fov_dev(st._h, C.byref(sp), C.byref(fp), L.ptr(rows), len(rows), C.byref(nr), C.byref(ns), C.byref(ni)))
print(nr.value, ns.value, ni.value)
