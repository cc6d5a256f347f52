"""Stand-alone timing of the two Gaussian filters of get_seeds on a resident 2048x2048x50 stack (developer tool)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis3_amd import synth, _lib as L
lib = L.lib(); L.check(lib.ia3_init(0))
dtype = np.uint16 if (len(sys.argv) > 1 and sys.argv[1] == "u16") else np.float32
im, c, h = synth.make_fov((50, 2048, 2048), 5000, 3, dtype=dtype)
a = L.DeviceStack.upload(im); b = L.DeviceStack.empty(im.shape, im.dtype)
for var in (0,):
  for sigma, trunc, mode in ((0.75, 4.0, L.MODE_REFLECT), (7.5, 4.0, L.MODE_REFLECT), (3.0, 2.0, L.MODE_NEAREST)):
    w, r = L.gaussian_taps(sigma, trunc)
    for rep in range(3):
        L.check(lib.ia3_gaussian_filter_dev(a._h, C.c_double(sigma), C.c_double(trunc), mode, L.dptr(w), r, b._h))
    L.check(lib.ia3_sync()); L.profile_enable(True)
    for rep in range(10):
        L.check(lib.ia3_gaussian_filter_dev(a._h, C.c_double(sigma), C.c_double(trunc), mode, L.dptr(w), r, b._h))
    prof = L.profile_collect(); L.profile_enable(False)
    print("var", var, np.dtype(dtype).name, "sigma", sigma, "R", r, {k: round(v[1] / v[0], 3) for k, v in prof.items() if k.startswith("gauss")})
  # the whole seeding stage with this variant (front filter overlapped with the first long pass)
  sp, keep = L.make_seed_params(600.0, max_num_seeds=None)
  out = np.empty((65536, 4)); nn = C.c_int(0); th = C.c_double(0)
  import time
  for rep in range(3):
      L.check(lib.ia3_dog_seed_dev(a._h, C.byref(sp), L.dptr(out), len(out), C.byref(nn), C.byref(th)))
  t0 = time.perf_counter()
  for rep in range(10):
      L.check(lib.ia3_dog_seed_dev(a._h, C.byref(sp), L.dptr(out), len(out), C.byref(nn), C.byref(th)))
  print("var", var, "dog_seed %.3f ms" % ((time.perf_counter() - t0) * 100), nn.value)
  L.profile_enable(True)
  for rep in range(5):
      L.check(lib.ia3_dog_seed_dev(a._h, C.byref(sp), L.dptr(out), len(out), C.byref(nn), C.byref(th)))
  prof = L.profile_collect(); L.profile_enable(False)
  print("   ", {k: round(v[1] / v[0], 3) for k, v in prof.items()})
