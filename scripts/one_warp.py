"""Two cubic warps of a 50x2048x2048 uint16 stack (developer tool for rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from imageanalysis3_amd import synth, _lib as L
L.check(L.lib().ia3_init(0))
rng = np.random.RandomState(1)
im = rng.randint(300, 900, size=(50, 2048, 2048)).astype(np.uint16)
a = L.DeviceStack.upload(im); b = L.DeviceStack.empty(im.shape, np.uint16)
d = np.array([0.6, -3.4, 5.2])
for _ in range(2):
    L.check(L.lib().ia3_warp3d_dev(a._h, L.dptr(d), None, 0, 3, L.MODE_NEAREST, C.c_double(0.0), b._h))
L.check(L.lib().ia3_sync())
print(b.download()[25, 1000, 1000])
