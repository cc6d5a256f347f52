"""Latency of LM iterations for a wave that runs alone (developer tool): a handful of noise-only balls fitted with
GaussianFit's defaults; reports microseconds per model evaluation."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis3_amd import _lib as L
from imageanalysis3_amd.External.Fitting_v4 import gaussfit_batch
L.check(L.lib().ia3_init(0))
r = 5
g = np.indices((2 * r + 1,) * 3).reshape(3, -1) - r
ball = g[:, (g ** 2).sum(0) <= r * r][:, :512]
n_fits = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rng = np.random.RandomState(7)
ims, Xs, cens = [], [], []
for i in range(n_fits):
    c = np.array([20, 100, 100])
    ims.append((400 + 15 * rng.randn(ball.shape[1])).astype(np.float32))
    Xs.append(ball + c[:, None]); cens.append(c)
for rep in range(3):
    L.check(L.lib().ia3_sync()); t0 = time.perf_counter()
    ps, xs, ok, nfev = gaussfit_batch(ims, Xs, np.array(cens))
    dt = time.perf_counter() - t0
    print("fits %d  max nfev %d (%d fits at maxfev)  wall %.2f ms  -> %.2f us per evaluation of the slowest fit" % (n_fits, nfev.max(), int((nfev >= 1000).sum()), dt * 1e3, dt * 1e6 / max(nfev.max(), 1)))
