"""Time align_image (phase-correlation path and bead path) and warp on a full-size pair (developer tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from imageanalysis3_amd import synth, _lib as L
from imageanalysis3_amd.correction_tools.alignment import align_image
from imageanalysis3_amd.correction_tools.translate import warp_3d_image
shape = (50, 2048, 2048)
t0 = time.time()
ref, src, c, h = synth.make_bead_pair(shape, 400, 21, (0.6, -3.4, 5.2), dtype=np.uint16)
print("gen %.1fs" % (time.time() - t0))
L.check(L.lib().ia3_init(0))
for use_autocorr in (True, False):
    for it in range(2):
        L.profile_enable(True); L.profile_collect()
        t0 = time.perf_counter()
        d, flag = align_image(src, ref, use_autocorr=use_autocorr, verbose=False,
                              correction_args={'single_im_size': shape})
        dt = time.perf_counter() - t0
        prof = L.profile_collect(); L.profile_enable(False)
        print("autocorr=%s run %d: %.1f ms drift %s flag %d" % (use_autocorr, it, dt * 1e3, np.round(d, 3), flag))
        if it == 1:
            print("   kernels:", {k: round(v[1], 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:8]})
a, b = L.DeviceStack.upload(src), L.DeviceStack.upload(ref)
for use_autocorr in (True, False):
    for it in range(2):
        t0 = time.perf_counter()
        d, flag = align_image(a, b, use_autocorr=use_autocorr, verbose=False, correction_args={'single_im_size': shape})
        dt = time.perf_counter() - t0
    print("resident autocorr=%s: %.1f ms drift %s" % (use_autocorr, dt * 1e3, np.round(d, 3)))
a.free(); b.free()
for order, mode in ((1, 'constant'), (3, 'nearest')):
    for it in range(2):
        L.profile_enable(True); L.profile_collect()
        t0 = time.perf_counter()
        out = warp_3d_image(src, d, warp_order=order, border_mode=mode)
        dt = time.perf_counter() - t0
        prof = L.profile_collect(); L.profile_enable(False)
    print("warp order %d: %.1f ms end to end; kernels: %s" % (order, dt * 1e3, {k: round(v[1], 2) for k, v in prof.items()}))
