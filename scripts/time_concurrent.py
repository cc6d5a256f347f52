"""Throughput of N resident FOVs processed by 1/2/4 host threads (developer tool): uint16 fields have a few fits that
run to maxfev, whose tail dominates a single stream."""
import ctypes as C, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
from imageanalysis3_amd import synth, _lib as L
lib = L.lib(); L.check(lib.ia3_init(0))
dtype = np.uint16 if (len(sys.argv) < 2 or sys.argv[1] == "u16") else np.float32
stacks = []
for k in range(4):
    im, c, h = synth.make_fov((50, 2048, 2048), 5000, 3 + k, dtype=dtype)
    stacks.append(L.DeviceStack.upload(im))
sp, keep = L.make_seed_params(600.0, max_num_seeds=None); fp = L.make_fit_params()

def one(st):
    rows = np.empty((65536, 11), np.float32); nr, ns, ni = C.c_int(0), C.c_int(0), C.c_int(0)
    L.check(lib.ia3_fit_fov_dev(st._h, C.byref(sp), C.byref(fp), L.ptr(rows), len(rows), C.byref(nr), C.byref(ns), C.byref(ni)))
    return nr.value

jobs = stacks * 4   # 16 FOV passes
for w in (1, 2, 4, 8, 16):
    t0 = time.perf_counter()
    if w == 1:
        res = [one(s) for s in jobs]
    else:
        with ThreadPoolExecutor(w) as pool:
            res = list(pool.map(one, jobs))
    dt = time.perf_counter() - t0
    print("%s: %d threads: %.1f ms per FOV (%d FOVs, %d rows)" % (np.dtype(dtype).name, w, dt / len(jobs) * 1e3, len(jobs), sum(res)))
