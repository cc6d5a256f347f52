import sys, os, time, ctypes as C, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'oracle'))
import np_oracle as O
from imageanalysis3_amd import synth, _lib as L
from imageanalysis3_amd.External.Fitting_v4 import iter_fit_seed_points
from imageanalysis3_amd.spot_tools.fitting import get_seeds
shape=(50,512,512); n=400
im,c,h = synth.make_fov(shape, n, 3, layout="clustered", n_territories=16)
seeds = get_seeds(im, th_seed=600.0)
print("seeds", len(seeds), flush=True)
fo = O.iter_fit_seed_points(im, seeds.T, voronoi="lowest_index"); fo.firstfit(); fo.repeatfit(); po=np.array(fo.ps,dtype=np.float64)
print("oracle n_iter", fo.n_iter, flush=True)
for rep in range(3):
    f = iter_fit_seed_points(im, seeds.T); f.firstfit(); f.repeatfit(); p=np.array(f.ps,dtype=np.float64)
    with np.errstate(all="ignore"):
        rel=np.abs(p[:,:8]-po[:,:8])/np.abs(po[:,:8])
    print(rep, "gpu n_iter", f.n_iter, "identical %d/%d max rel %.2e" % ((np.nanmax(rel,1)==0).sum(), len(p), np.nanmax(rel)), flush=True)
