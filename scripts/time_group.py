"""Group fit of a few resident uint16 FOVs against one call per FOV (developer tool)."""
import ctypes as C, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis3_amd import synth, _lib as L
lib = L.lib(); L.check(lib.ia3_init(0))
shape = (50, 2048, 2048)
ims = [synth.make_fov(shape, 5000, 40 + k, dtype=np.uint16)[0] for k in range(3)]
sts = [L.DeviceStack.upload(im) for im in ims]
sp, keep = L.make_seed_params(600.0, max_num_seeds=None); fp = L.make_fit_params()
def T(): lib.ia3_sync(); return time.perf_counter()
for rep in range(3):
    t0 = T()
    one = [L.fit_fovs([st], sp, fp, in_flight=1) for st in sts]
    t1 = T()
    tabs, info = L.fit_fovs(sts, sp, fp, in_flight=3)
    t2 = T()
    print("one by one %.1f ms (%s); group of 3 %.1f ms (%s)" % (
        1e3 * (t1 - t0), [(o[1][0]["n_iter"], o[1][0]["nfev"]) for o in one], 1e3 * (t2 - t1), [(i["n_iter"], i["nfev"]) for i in info]), flush=True)
    assert all(np.array_equal(a, o[0][0]) for a, o in zip(tabs, one))
