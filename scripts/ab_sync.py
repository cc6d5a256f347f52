"""A/B of IA3_TUNE_SYNC_SEEDS on one box: float32 single stream, uint16 batched (developer probe)."""
import sys, os, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch; torch.cuda.set_device(0); torch.zeros(1, device="cuda")
from imageanalysis3_amd import synth, _lib as L
lib = L.lib(); L.check(lib.ia3_init(0))
f32 = synth.make_fov((50, 2048, 2048), 5000, 3)[0]
u16 = synth.make_fov((50, 2048, 2048), 5000, 40, dtype=np.uint16)[0]
sp, keep = L.make_seed_params(600.0, max_num_seeds=None); fp = L.make_fit_params()
sf, su = L.DeviceStack.upload(f32), L.DeviceStack.upload(u16)
rows = np.empty((16384, 11), np.float32); nr, ns, ni = C.c_int(0), C.c_int(0), C.c_int(0)
for rep in range(3):
    for mode in (1, 0):
        L.check(lib.ia3_set_tuning(7, mode))
        for _ in range(3):
            L.check(lib.ia3_fit_fov_dev(sf._h, C.byref(sp), C.byref(fp), L.ptr(rows), len(rows), C.byref(nr), C.byref(ns), C.byref(ni)))
        lib.ia3_sync(); t0 = time.perf_counter()
        for _ in range(20):
            L.check(lib.ia3_fit_fov_dev(sf._h, C.byref(sp), C.byref(fp), L.ptr(rows), len(rows), C.byref(nr), C.byref(ns), C.byref(ni)))
        dt = (time.perf_counter() - t0) / 20
        L.fit_fovs([su] * 12, sp, fp, in_flight=12)
        t0 = time.perf_counter()
        L.fit_fovs([su] * 24, sp, fp, in_flight=12)
        du = (time.perf_counter() - t0) / 24
        print("sync_seeds=%d: f32 one stream %.3f ms/FOV (%d rows), u16 batched(12) %.2f ms/FOV" % (mode, dt * 1e3, nr.value, du * 1e3), flush=True)
