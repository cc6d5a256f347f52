"""Stress of ia3_fit_fovs (developer tool): several caller threads submit batches of different shapes and dtypes at the
same time, mixing host and resident jobs, repeatedly; every table must equal the one a lone sequential call gives."""
import os, sys, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis3_amd import synth, _lib as L
lib = L.lib(); L.check(lib.ia3_init(0))
sp, keep = L.make_seed_params(600.0, max_num_seeds=None); fp = L.make_fit_params()
specs = [((30, 128, 128), 50, np.float32, "isolated"), ((30, 128, 128), 50, np.uint16, "isolated"),
         ((50, 256, 256), 120, np.float32, "clustered"), ((24, 200, 260), 60, np.uint16, "isolated"),
         ((50, 512, 512), 400, np.float32, "clustered")]
groups = []
for k, (shape, n, dt, layout) in enumerate(specs):
    ims = [synth.make_fov(shape, n, 10 * k + j, dtype=dt, layout=layout, **({"n_territories": 12} if layout == "clustered" else {}))[0]
           for j in range(3)]
    ref, _ = L.fit_fovs(ims, sp, fp, in_flight=1)
    groups.append((ims, ref))
print("references made", [len(r[0]) for _, r in groups], flush=True)
errors = []

def worker(tid, reps):
    try:
        for rep in range(reps):
            ims, ref = groups[(tid + rep) % len(groups)]
            stacks = [L.DeviceStack.upload(ims[0])] if rep % 2 else []
            jobs = (stacks + ims[1:]) if stacks else list(ims)
            tabs, info = L.fit_fovs(jobs * 2, sp, fp, in_flight=1 + (tid + rep) % 4)
            for a, b in zip(tabs, list(ref) * 2):
                if a.shape != b.shape or not np.array_equal(a, b):
                    errors.append((tid, rep, a.shape, b.shape))
            for s in stacks:
                s.free()
    except Exception as e:   # noqa
        errors.append((tid, repr(e)))

threads = [threading.Thread(target=worker, args=(t, 12)) for t in range(5)]
for t in threads: t.start()
for t in threads: t.join()
print("errors:", errors[:5])
assert not errors
print("stress ok")
