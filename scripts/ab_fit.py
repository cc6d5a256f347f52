"""A/B of the fit kernel (developer tool): one resident bench FOV, seed list fixed, ia3_fit_run timed with HIP events
through the library's per-kernel profile for IA3_TUNE_FIT_WAVES = 1 / 2; tables compared bit for bit.
usage: python scripts/ab_fit.py [Z X Y n layout dtype]"""
import ctypes as C, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis3_amd import synth, _lib as L
lib = L.lib(); L.check(lib.ia3_init(0))
shape = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (50, 2048, 2048)
n = int(sys.argv[4]) if len(sys.argv) > 4 else 5000
layout = sys.argv[5] if len(sys.argv) > 5 else "isolated"
dtype = sys.argv[6] if len(sys.argv) > 6 else "float32"
im, c, h = synth.make_fov(shape, n, 3, layout=layout)
if dtype == "uint16":
    im = np.clip(np.rint(im), 0, 65535).astype(np.uint16)
st = L.DeviceStack.upload(im)
sp, keep = L.make_seed_params(600.0, max_num_seeds=None); fp = L.make_fit_params()
out = np.empty((65536, 4)); nn = C.c_int(0); th = C.c_double(0)
L.check(lib.ia3_dog_seed_dev(st._h, C.byref(sp), L.dptr(out), len(out), C.byref(nn), C.byref(th)))
seeds = np.ascontiguousarray(out[:nn.value, :3])
def T(): lib.ia3_sync(); return time.perf_counter()
tables = {}
for waves in (1, 2, 1, 2):
    L.check(lib.ia3_set_tuning(10, waves))
    best = 1e9
    for rep in range(4):
        hh = C.c_void_p()
        L.check(lib.ia3_fit_create(st._h, L.dptr(seeds), len(seeds), C.byref(fp), C.byref(hh)))
        t0 = T(); L.check(lib.ia3_fit_run(hh)); t1 = T()
        ps = np.empty((len(seeds), 11), np.float32); it = C.c_int(0)
        L.check(lib.ia3_fit_results_ex(hh, L.ptr(ps), None, None, C.byref(it)))
        a, b = C.c_int64(0), C.c_int64(0); lib.ia3_fit_stats(hh, C.byref(a), C.byref(b))
        lib.ia3_fit_destroy(hh)
        best = min(best, 1e3 * (t1 - t0))
    tables.setdefault(waves, ps)
    print("waves/SIMD %d: seeds %d sweeps %d fits %d nfev %d  fit_run best %.3f ms" % (waves, len(seeds), it.value, a.value, b.value, best), flush=True)
same = np.array_equal(tables[1], tables[2], equal_nan=True)
print("tables identical:", same)
# fixed vs per-evaluation part (IA3_DEBUG_FIT_MAXFEV: results differ, timing only)
if os.environ.get("AB_MAXFEV", "1") == "1":
    L.check(lib.ia3_set_tuning(10, 2))
    for mf in (1, 2, 3, 4, 6, 0):
        L.check(lib.ia3_set_tuning(100, mf))
        best = 1e9
        for rep in range(3):
            hh = C.c_void_p()
            L.check(lib.ia3_fit_create(st._h, L.dptr(seeds), len(seeds), C.byref(fp), C.byref(hh)))
            t0 = T(); L.check(lib.ia3_fit_run(hh)); t1 = T()
            a, b = C.c_int64(0), C.c_int64(0); lib.ia3_fit_stats(hh, C.byref(a), C.byref(b))
            lib.ia3_fit_destroy(hh)
            best = min(best, 1e3 * (t1 - t0))
        print("maxfev %d: fits %d nfev %d  fit_run best %.3f ms" % (mf, a.value, b.value, best), flush=True)
    L.check(lib.ia3_set_tuning(100, 0))
