"""Per-kernel HIP-event times of ia3_dog_filters_dev on a bench-size stack, nothing else on the chip (developer probe)."""
import sys, os, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis3_amd import synth, _lib as L
lib = L.lib(); L.check(lib.ia3_init(0))
for dt in (np.float32, np.uint16):
    im = synth.make_fov((50, 2048, 2048), 5000, 3, dtype=dt)[0]
    st = L.DeviceStack.upload(im); f = L.DeviceStack.empty(im.shape, dt); b = L.DeviceStack.empty(im.shape, dt)
    for fold in (1, 0):
        L.check(lib.ia3_set_tuning(8, fold))
        for _ in range(3):
            L.check(lib.ia3_dog_filters_dev(st._h, C.c_double(0.75), C.c_double(7.5), f._h, b._h))
        L.profile_enable(True); L.profile_collect()
        lib.ia3_sync(); t0 = time.perf_counter()
        for _ in range(10):
            L.check(lib.ia3_dog_filters_dev(st._h, C.c_double(0.75), C.c_double(7.5), f._h, b._h))
        lib.ia3_sync(); d = (time.perf_counter() - t0) / 10
        prof = L.profile_collect(); L.profile_enable(False)
        print("%s fold=%d: %.3f ms, %s" % (np.dtype(dt).name, fold, d * 1e3, {k: round(v[1] / v[0], 3) for k, v in prof.items()}), flush=True)
    st.free(); f.free(); b.free()
L.check(lib.ia3_set_tuning(8, 1))
