"""IA3_TUNE_FIT_MERGE on / off (developer tool): a lone uint16 FOV, a group of three uint16 FOVs (the movie leg's fit), a
crowded float32 FOV; wall time per call and identical tables."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis3_amd import synth, _lib as L
lib = L.lib(); L.check(lib.ia3_init(0))
shape = (50, 2048, 2048)
u16 = [synth.make_fov(shape, 5000, 40 + i, dtype=np.uint16)[0] for i in range(3)]
clu = synth.make_fov(shape, 5000, 7, layout="clustered", n_territories=200)[0]
sp, keep = L.make_seed_params(600.0, max_num_seeds=None); fp = L.make_fit_params()
stacks = [L.DeviceStack.upload(a) for a in u16]
sclu = L.DeviceStack.upload(clu)
cases = {"lone uint16": ([stacks[0]], 1), "three uint16 (group)": (stacks, 3), "crowded float32": ([sclu], 1),
         "three uint16 + crowded? no: twelve uint16 (group)": (stacks * 4, 12)}
res = {}
for merge in (1, 0, 1, 0):
    L.check(lib.ia3_set_tuning(11, merge))
    for name, (ims, depth) in cases.items():
        L.fit_fovs(ims, sp, fp, in_flight=depth)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); tabs, info = L.fit_fovs(ims, sp, fp, in_flight=depth); ts.append(time.perf_counter() - t0)
        key = (name, merge)
        if key in res:
            assert all(np.array_equal(a, b) for a, b in zip(res[key][1], tabs))
        res[key] = (min(ts), tabs, [i["n_iter"] for i in info])
        print("merge %d  %-28s best %.2f ms  sweeps %s rows %s" % (merge, name[:28], min(ts) * 1e3, res[key][2][:3], [len(t) for t in tabs][:3]), flush=True)
for name in cases:
    a, b = res[(name, 1)][1], res[(name, 0)][1]
    print(name[:28], "tables identical:", all(np.array_equal(x, y) for x, y in zip(a, b)))
