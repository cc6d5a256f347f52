"""A/B of IA3_TUNE_GAUSS_FOLD (8) on one box: whole FOV from one stream + per-kernel HIP-event times (developer probe)."""
import sys, os, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis3_amd import synth, _lib as L
lib = L.lib(); L.check(lib.ia3_init(0))
for dt in (np.float32, np.uint16):
    im = synth.make_fov((50, 2048, 2048), 5000, 3, dtype=dt)[0]
    sp, keep = L.make_seed_params(600.0, max_num_seeds=None); fp = L.make_fit_params()
    st = L.DeviceStack.upload(im)
    rows = np.empty((16384, 11), np.float32); nr, ns, ni = C.c_int(0), C.c_int(0), C.c_int(0)
    tables = {}
    for rep in range(2):
        for mode in (0, 1):
            L.check(lib.ia3_set_tuning(8, mode))
            for _ in range(2):
                L.check(lib.ia3_fit_fov_dev(st._h, C.byref(sp), C.byref(fp), L.ptr(rows), len(rows), C.byref(nr), C.byref(ns), C.byref(ni)))
            tables[mode] = rows[:nr.value].copy()
            L.profile_enable(True); L.profile_collect()
            lib.ia3_sync(); t0 = time.perf_counter()
            for _ in range(10):
                L.check(lib.ia3_fit_fov_dev(st._h, C.byref(sp), C.byref(fp), L.ptr(rows), len(rows), C.byref(nr), C.byref(ns), C.byref(ni)))
            d = (time.perf_counter() - t0) / 10
            prof = L.profile_collect(); L.profile_enable(False)
            print("%s fold=%d: %.3f ms/FOV, %d rows, %s" % (np.dtype(dt).name, mode, d * 1e3, nr.value,
                  {k: round(v[1] / v[0], 3) for k, v in prof.items() if "gauss" in k or "seed" in k}), flush=True)
    print("tables identical:", np.array_equal(tables[0].view(np.uint32), tables[1].view(np.uint32)), flush=True)
    st.free()
L.check(lib.ia3_set_tuning(8, 1))
