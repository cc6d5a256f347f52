"""Fit kernel timings on the bench FOV for the library named by IA3_LIB_PATH (default: the shipped one): fused / unfused
work lists, 1..3 persistent waves per SIMD (as far as the build allows); a CRC of each table.
usage: python scripts/ab_fit2.py [n_spots]"""
import ctypes as C, sys, time, os, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis3_amd import synth, _lib as L
lib = L.lib(); L.check(lib.ia3_init(0))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
im, c, h = synth.make_fov((50, 2048, 2048), n, 3)
st = L.DeviceStack.upload(im)
sp, keep = L.make_seed_params(600.0, max_num_seeds=None); fp = L.make_fit_params()
out = np.empty((65536, 4)); nn = C.c_int(0); th = C.c_double(0)
L.check(lib.ia3_dog_seed_dev(st._h, C.byref(sp), L.dptr(out), len(out), C.byref(nn), C.byref(th)))
seeds = np.ascontiguousarray(out[:nn.value, :3])
print("library", L.LIB_PATH, "seeds", len(seeds), flush=True)
def T(): lib.ia3_sync(); return time.perf_counter()
for fuse in (1, 0):
    for waves in (2, 3):
        L.check(lib.ia3_set_tuning(7, fuse)); L.check(lib.ia3_set_tuning(10, waves))
        best, ts = 1e9, []
        for rep in range(6):
            hh = C.c_void_p()
            L.check(lib.ia3_fit_create(st._h, L.dptr(seeds), len(seeds), C.byref(fp), C.byref(hh)))
            t0 = T(); L.check(lib.ia3_fit_run(hh)); t1 = T()
            ps = np.empty((len(seeds), 11), np.float32); it = C.c_int(0)
            L.check(lib.ia3_fit_results_ex(hh, L.ptr(ps), None, None, C.byref(it)))
            a, b = C.c_int64(0), C.c_int64(0); lib.ia3_fit_stats(hh, C.byref(a), C.byref(b))
            lib.ia3_fit_destroy(hh)
            ts.append(1e3 * (t1 - t0))
        print("fuse %d waves/SIMD %d (asked): fits %d nfev %d  fit_run min %.3f median %.3f ms  table crc %08x" % (
            fuse, waves, a.value, b.value, min(ts), sorted(ts)[len(ts) // 2], zlib.crc32(ps.tobytes())), flush=True)
L.check(lib.ia3_set_tuning(7, 1)); L.check(lib.ia3_set_tuning(10, 2))
