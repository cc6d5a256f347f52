"""Which fits of a uint16 FOV run to maxfev (developer tool)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis3_amd import synth, _lib as L
lib = L.lib(); L.check(lib.ia3_init(0))
dtype = np.float32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else np.uint16
im, c, h = synth.make_fov((50, 2048, 2048), 5000, 3, dtype=dtype)
st = L.DeviceStack.upload(im)
sp, keep = L.make_seed_params(600.0, max_num_seeds=None); fp = L.make_fit_params()
out = np.empty((65536, 4)); nn = C.c_int(0); th = C.c_double(0)
L.check(lib.ia3_dog_seed_dev(st._h, C.byref(sp), L.dptr(out), len(out), C.byref(nn), C.byref(th)))
seeds = np.ascontiguousarray(out[:nn.value, :3]); hs = out[:nn.value, 3]
print("seeds", len(seeds), "th", th.value)
hh = C.c_void_p()
L.check(lib.ia3_fit_create(st._h, L.dptr(seeds), len(seeds), C.byref(fp), C.byref(hh)))
L.check(lib.ia3_fit_first(hh))
nf = np.empty(len(seeds), np.int32); L.check(lib.ia3_fit_nfev(hh, L.ptr(nf)))
ps = np.empty((len(seeds), 11), np.float32); L.check(lib.ia3_fit_results(hh, L.ptr(ps), None, None))
print("first fit: nfev histogram", np.bincount(np.minimum(nf // 100, 10)))
from scipy.spatial import cKDTree
d, j = cKDTree(c).query(seeds)
for i in np.where(nf >= 900)[0]:
    z, x, y = seeds[i].astype(int)
    print("seed", seeds[i], "h_seed %.0f" % hs[i], "dist to injected %.2f" % d[i], "nfev", nf[i], "row", np.round(ps[i], 3))
    print("   centre column z profile:", im[max(z - 3, 0):z + 4, x, y].astype(int), " plane row:", im[z, x, max(y - 3, 0):y + 4].astype(int))
ni = C.c_int(0)
L.check(lib.ia3_fit_repeat(hh, C.byref(ni)))
nf2 = np.empty(len(seeds), np.int32); L.check(lib.ia3_fit_nfev(hh, L.ptr(nf2)))
ps2 = np.empty((len(seeds), 11), np.float32); L.check(lib.ia3_fit_results(hh, L.ptr(ps2), None, None))
print("sweeps", ni.value, "sweep nfev histogram", np.bincount(np.minimum((nf2 - nf) // 100, 10)))
tree = cKDTree(seeds)
dd, jj = tree.query(seeds, k=2)
for i in np.where(nf2 - nf >= 300)[0]:
    print("seed", seeds[i], "nfev first %d sweeps %d" % (nf[i], nf2[i] - nf[i]), "nearest other seed %.2f" % dd[i, 1], "dist to injected %.2f" % d[i])
    print("   first", np.round(ps[i], 3)); print("   final", np.round(ps2[i], 3))
    k = jj[i, 1]
    print("   neighbour", seeds[k], "first", np.round(ps[k, :8], 3), "final", np.round(ps2[k, :8], 3))
print("pairs closer than 4 px:", int((dd[:, 1] < 4).sum()), " of which slow:", int(((dd[:, 1] < 4) & (nf2 - nf >= 300)).sum()))
