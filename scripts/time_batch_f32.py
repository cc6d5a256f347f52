"""ia3_fit_fovs on resident float32 bench FOVs: ms per FOV against the group size (developer tool)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis3_amd import synth, _lib as L
lib = L.lib(); L.check(lib.ia3_init(0))
shape = (50, 2048, 2048)
sts = [L.DeviceStack.upload(synth.make_fov(shape, 5000, 40 + k)[0]) for k in range(4)]
for a in sys.argv[1:]:
    if a.startswith("waves="):
        L.check(lib.ia3_set_tuning(10, int(a[6:])))   # IA3_TUNE_FIT_WAVES
sp, keep = L.make_seed_params(600.0, max_num_seeds=None); fp = L.make_fit_params()
for g in (8, 32):
    ims = [sts[k % 4] for k in range(64)]
    L.fit_fovs(ims[:2 * g], sp, fp, in_flight=g)
    lib.ia3_sync(); t0 = time.perf_counter()
    tabs, info = L.fit_fovs(ims, sp, fp, in_flight=g)
    dt = time.perf_counter() - t0
    print("in_flight %2d: %.3f ms per FOV, %.2f M spots/s" % (g, dt / 64 * 1e3, sum(len(t) for t in tabs) / dt / 1e6), flush=True)
