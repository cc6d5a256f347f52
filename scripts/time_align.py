"""align_image on a resident full-size uint16 bead pair: plain reference stack against a DriftReference (crop spectra kept);
ms per call and the kernel time of its two parts (developer tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from imageanalysis3_amd import synth, _lib as L
from imageanalysis3_amd.correction_tools.alignment import align_image, DriftReference
shape = (50, 2048, 2048)
ref, src, c, h = synth.make_bead_pair(shape, 300, 43, (0.6, -3.4, 5.2), dtype=np.uint16)
lib = L.lib(); L.check(lib.ia3_init(0))
a, b = L.DeviceStack.upload(src), L.DeviceStack.upload(ref)
kw = dict(use_autocorr=True, verbose=False, correction_args={'single_im_size': shape})
dref = DriftReference(b)
for name, r in (("plain reference stack", b), ("DriftReference", dref)):
    align_image(a, r, **kw)
    L.check(lib.ia3_sync()); t0 = time.perf_counter()
    for _ in range(20):
        d, f = align_image(a, r, **kw)
    dt = (time.perf_counter() - t0) / 20
    L.profile_enable(True); L.profile_collect()
    for _ in range(5):
        align_image(a, r, **kw)
    L.check(lib.ia3_sync()); prof = L.profile_collect(); L.profile_enable(False)
    print("%s: %.3f ms per align_image, drift %s flag %d; kernels per call: %s" % (
        name, dt * 1e3, np.round(d, 3), f, {k: round(v[1] / 5, 3) for k, v in prof.items()}), flush=True)
dref.free(); a.free(); b.free()
