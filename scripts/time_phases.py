"""Host-side phase timing of one FOV through the C ABI (developer tool)."""
import ctypes as C, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis3_amd import synth, _lib as L
lib = L.lib(); L.check(lib.ia3_init(0))
shape = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (50, 2048, 2048)
n = int(sys.argv[4]) if len(sys.argv) > 4 else 5000
layout = sys.argv[5] if len(sys.argv) > 5 else "isolated"
im, c, h = synth.make_fov(shape, n, 3, layout=layout)
st = L.DeviceStack.upload(im)
sp, keep = L.make_seed_params(600.0, max_num_seeds=None); fp = L.make_fit_params()
def T(): lib.ia3_sync(); return time.perf_counter()
for rep in range(3):
    out = np.empty((65536, 4)); nn = C.c_int(0); th = C.c_double(0)
    t0 = T(); L.check(lib.ia3_dog_seed_dev(st._h, C.byref(sp), L.dptr(out), len(out), C.byref(nn), C.byref(th))); t1 = T()
    seeds = np.ascontiguousarray(out[:nn.value, :3]); hh = C.c_void_p()
    L.check(lib.ia3_fit_create(st._h, L.dptr(seeds), len(seeds), C.byref(fp), C.byref(hh))); t2 = T()
    L.check(lib.ia3_fit_first(hh)); t3 = T()
    L.check(lib.ia3_fit_repeat(hh, None)); t4 = T()
    ps = np.empty((len(seeds), 11), np.float32); it = C.c_int(0)
    L.check(lib.ia3_fit_results_ex(hh, L.ptr(ps), None, None, C.byref(it))); t5 = T()
    a, b = C.c_int64(0), C.c_int64(0); lib.ia3_fit_stats(hh, C.byref(a), C.byref(b))
    lib.ia3_fit_destroy(hh); t6 = T()
    rows = np.empty((65536, 11), np.float32); nr, ns, ni = C.c_int(0), C.c_int(0), C.c_int(0)
    L.check(lib.ia3_fit_fov_dev(st._h, C.byref(sp), C.byref(fp), L.ptr(rows), len(rows), C.byref(nr), C.byref(ns), C.byref(ni))); t7 = T()
    print("seeds %d sweeps %d fits %d nfev %d | seed %.2f create %.2f first %.2f repeat %.2f results %.2f destroy %.2f | fused %.2f ms"
          % (nn.value, it.value, a.value, b.value, *(1e3 * (y - x) for x, y in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t5, t6), (t6, t7)))))
