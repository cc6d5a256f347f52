"""uint16 FOV through ia3_fit_fovs at several in-flight depths (developer probe; GPU_MAX_HW_QUEUES from the environment)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis3_amd import synth, _lib as L
lib = L.lib(); L.check(lib.ia3_init(0))
u16, c, h = synth.make_fov((50, 2048, 2048), 5000, 40, dtype=np.uint16)
sp, keep = L.make_seed_params(600.0, max_num_seeds=None); fp = L.make_fit_params()
st = L.DeviceStack.upload(u16)
print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"), flush=True)
for depth in [int(x) for x in sys.argv[1:]] or [8, 12, 16]:
    L.fit_fovs([st] * depth, sp, fp, in_flight=depth)
    n = 2 * depth if depth > 8 else 24
    t0 = time.perf_counter()
    tabs, info = L.fit_fovs([st] * n, sp, fp, in_flight=depth)
    dt = time.perf_counter() - t0
    print("depth %2d: %.2f ms/FOV over %d FOVs" % (depth, dt / n * 1e3, n), flush=True)
