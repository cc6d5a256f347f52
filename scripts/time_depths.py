"""get_seeds stage time per voxel for several stack depths (developer tool): built depths take the column kernel,
others the sliding-window kernels."""
import ctypes as C, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis3_amd import synth, _lib as L
lib = L.lib(); L.check(lib.ia3_init(0))
sp, keep = L.make_seed_params(600.0, max_num_seeds=None)
for Z in (25, 33, 35, 45, 48, 50, 60):
    im, c, h = synth.make_fov((Z, 2048, 2048), 100 * Z, 3)
    st = L.DeviceStack.upload(im)
    out = np.empty((65536, 4)); nn = C.c_int(0); th = C.c_double(0)
    best = 1e9
    for rep in range(4):
        lib.ia3_sync(); t0 = time.perf_counter()
        L.check(lib.ia3_dog_seed_dev(st._h, C.byref(sp), L.dptr(out), len(out), C.byref(nn), C.byref(th)))
        lib.ia3_sync(); best = min(best, time.perf_counter() - t0)
    st.free()
    print("Z %d: get_seeds %.3f ms, %.3f ns per voxel, %d seeds" % (Z, best * 1e3, best * 1e9 / im.size, nn.value), flush=True)
