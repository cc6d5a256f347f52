"""Per-phase shader cycles of the fit kernel (profiling build, scripts/fit_stamps.sh).
usage: fit_stamps.py [Z X Y n layout dtype]"""
import ctypes as C, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imageanalysis3_amd import _lib as L
L.LIB_PATH = os.path.join(ROOT, "imageanalysis3_amd", "libia3_stamps.so")
from imageanalysis3_amd import synth
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
names = {0: "ticket", 1: "gather", 2: "extremes", 3: "fit set-up", 4: "solver tail", 6: "geometry", 7: "voxel slots",
         8: "cross-lane sums", 9: "algebra", 10: "natural+eps", 11: "store_result", 12: "hand-over", 13: "admission", 14: "store drain"}
for waves in (1, 2):
    L.check(lib.ia3_set_tuning(10, waves))
    for rep in range(2):
        hh = C.c_void_p()
        L.check(lib.ia3_fit_create(st._h, L.dptr(seeds), len(seeds), C.byref(fp), C.byref(hh)))
        L.check(lib.ia3_fit_run(hh))
        cnt = np.zeros(32, np.uint64)
        L.check(lib.ia3_fit_counters(hh, cnt.ctypes.data_as(C.c_void_p)))
        lib.ia3_fit_destroy(hh)
    tot = float(cnt[8:].sum() - cnt[8 + 15] - cnt[8 + 16])
    nw = 4 * 256 * waves
    print("waves/SIMD %d: fits %d nfev %d; stamped cycles per wave %.0f (%.1f us at 2.4 GHz)" % (waves, cnt[0], cnt[1], tot / nw, tot / nw / 2400))
    if cnt[8 + 16]:
        print("  fits counted in the per-fit phases: %d with %d evaluations -> per evaluation: geometry %.0f, voxel slots %.0f, cross-lane sums %.0f, algebra %.0f cycles"
              % (cnt[8 + 16], cnt[8 + 15], cnt[8 + 6] / cnt[8 + 15], cnt[8 + 7] / cnt[8 + 15], cnt[8 + 8] / cnt[8 + 15], cnt[8 + 9] / cnt[8 + 15]))
    for k in range(24):
        if k in (15, 16):
            continue
        if cnt[8 + k]:
            print("  %-16s %6.2f %%  %9.0f cycles per fit" % (names.get(k, str(k)), 100 * float(cnt[8 + k]) / tot, float(cnt[8 + k]) / float(cnt[0])))
