#!/usr/bin/env python3
"""Cubic warp of a resident 50 x 2048 x 2048 stack with drift + dense chromatic field: wall time per call and the
per-kernel times of the library's profiler.  usage: time_warp.py [Z X Y] [uint16|float32]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from imageanalysis3_amd import _lib as L

a = [x for x in sys.argv[1:] if x.isdigit()]
Z, X, Y = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (50, 2048, 2048)
dt = np.float32 if "float32" in sys.argv else np.uint16
rng = np.random.default_rng(3)
im = rng.integers(100, 5000, size=(Z, X, Y)).astype(dt)
xx, yy = np.meshgrid(np.arange(X, dtype=np.float64), np.arange(Y, dtype=np.float64), indexing="ij")
smooth = np.stack([0.3 * np.sin(xx / 300.0) * np.cos(yy / 400.0), 0.8 * (xx / X - 0.5) ** 2 + 0.2 * (yy / Y - 0.5),
                   -0.6 * (yy / Y - 0.5) ** 2 + 0.1 * np.cos(xx / 500.0)])
field = np.ascontiguousarray(smooth[:, None].astype(np.float32) * np.ones((1, Z, 1, 1), np.float32))
drift = np.array([-0.6, 3.4, -5.17])
lib = L.lib()
for a_ in sys.argv[1:]:
    if a_.startswith("knob="):
        L.check(lib.ia3_set_tuning(12, int(a_[5:])))   # IA3_TUNE_WARP_ONEPASS
fp = C.c_void_p()
L.check(lib.ia3_buffer_upload(L.ptr(field), C.c_size_t(field.nbytes), C.byref(fp)))
with L.DeviceStack.upload(im) as src, L.DeviceStack.empty(im.shape, im.dtype) as dst:
    nofield = "nofield" in sys.argv
    def run():
        L.check(lib.ia3_warp3d_dev(src._h, L.dptr(drift), None if nofield else fp, 0 if nofield else 1, 3, L.MODE_NEAREST,
                                   C.c_double(0.0), dst._h))
        lib.ia3_sync()
    run()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); run(); ts.append(time.perf_counter() - t0)
    print("%s %dx%dx%d: warp order 3 + field best %.2f ms, median %.2f ms" % (np.dtype(dt).name, Z, X, Y, min(ts) * 1e3,
                                                                            sorted(ts)[2] * 1e3))
    L.profile_enable(True)
    for _ in range(3):
        run()
    for k, v in sorted(L.profile_collect().items()):
        print("  %-24s %s" % (k, v))
    L.profile_enable(False)
    out = dst.download()
    import zlib
    print("crc %08x" % zlib.crc32(out.tobytes()))
lib.ia3_buffer_free(fp)
