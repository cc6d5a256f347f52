"""Ten align_image(use_autocorr=True) calls on a RESIDENT 50x2048x2048 uint16 bead pair (developer tool for rocprofv3 and
for the wall time per call)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from imageanalysis3_amd import synth, _lib as L
from imageanalysis3_amd.correction_tools.alignment import align_image
shape = (50, 2048, 2048)
ref, src, c, h = synth.make_bead_pair(shape, 400, 21, (0.6, -3.4, 5.2), dtype=np.uint16)
L.check(L.lib().ia3_init(0))
a, b = L.DeviceStack.upload(src), L.DeviceStack.upload(ref)
ts = []
for _ in range(12):
    t0 = time.perf_counter()
    d, flag = align_image(a, b, use_autocorr=True, verbose=False, correction_args={'single_im_size': shape})
    ts.append(time.perf_counter() - t0)
print("drift", np.round(d, 4), "flag", flag, "ms per call: min %.2f median %.2f" % (min(ts[2:]) * 1e3, sorted(ts[2:])[5] * 1e3))
L.profile_enable(True); L.profile_collect()
for _ in range(5):
    align_image(a, b, use_autocorr=True, verbose=False, correction_args={'single_im_size': shape})
for k, v in sorted(L.profile_collect().items(), key=lambda kv: -kv[1][1]):
    print("  %-24s n=%d total %.3f ms (per call %.3f)" % (k, v[0], v[1], v[1] / 5))
a.free(); b.free()
