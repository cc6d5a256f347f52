"""configs[4] movie leg of bench.py on its own (developer tool): movies per second from 1, 2, 3 host threads, twice,
with the split between the correction chain and the fits.  IA3_LIB=<path> times another build of the library."""
import contextlib, io, json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis3_amd import synth, _lib as L
if os.environ.get("IA3_LIB"):
    L.LIB_PATH = os.environ["IA3_LIB"]
from imageanalysis3_amd.io_tools.load import correct_fov_image, DeviceBuffer

Z, X, Y = 50, 2048, 2048
chs = ['750', '647', '561', '488']
lib = L.lib()
L.check(lib.ia3_init(0))
ims = [synth.make_fov((Z, X, Y), 5000 if i < 3 else 300, 40 + i, dtype=np.uint16)[0] for i in range(4)]
raw = np.empty((4 * Z, X, Y), np.uint16)
for i in range(4):
    raw[i::4] = ims[i]
yy, xx = np.meshgrid(np.linspace(-1, 1, Y), np.linspace(-1, 1, X))
bump = (0.55 + 0.45 * np.exp(-(xx ** 2 + yy ** 2))).astype(np.float32)
illum = {c: DeviceBuffer(bump) for c in chs}
bleed = np.zeros((3, 3, X, Y), np.float32)
for p in range(3):
    for q in range(3):
        bleed[p, q] = 1.0 if p == q else 0.05
bleed = DeviceBuffer(bleed)
zz = np.linspace(-1, 1, Z, dtype=np.float32)[:, None, None]
field = np.stack([0.2 * zz + 0 * xx[None].astype(np.float32), (0.8 * xx[None] + 0 * zz).astype(np.float32),
                  (0.8 * yy[None] + 0 * zz).astype(np.float32)]).astype(np.float32)
chrom = {'750': DeviceBuffer(field), '647': None, '561': DeviceBuffer(-field)}
sp, _ = L.make_seed_params(600.0, max_num_seeds=None)
fp = L.make_fit_params()
dref = L.DeviceStack.upload(ims[3])


INFO = []
THREADS = tuple(int(x) for x in os.environ.get("C5_THREADS", "1,2,3").split(","))
ROWS = []   # rows per channel of every movie run (they must all be equal: same input)


def movie(_=None):
    t_a = time.perf_counter()
    out, drift, flag = correct_fov_image(raw, chs[:3], single_im_size=[Z, X, Y], all_channels=chs, num_buffer_frames=0,
                                         num_empty_frames=0, calculate_drift=True, ref_filename=dref, corr_channels=chs[:3],
                                         illumination_profile=illum, bleed_profile=bleed, chromatic_profile=chrom,
                                         warp_image=True, return_drift=True, return_device=True, verbose=True)
    t_b = time.perf_counter()
    try:
        tabs, info = L.fit_fovs(out, sp, fp, in_flight=3)
    finally:
        for s in out:
            s.free()
    ROWS.append(tuple(len(t) for t in tabs))
    INFO.append([(i["n_seeds"], i["n_iter"], i["fits"], i["nfev"]) for i in info])
    return t_b - t_a, time.perf_counter() - t_b, sum(len(t) for t in tabs)


with contextlib.redirect_stdout(io.StringIO()):
    movie()
    res = []
    for rep in range(2):
        for thr in THREADS:
            n = 2 * thr if thr > 1 else 3
            with ThreadPoolExecutor(max_workers=thr) as pool:
                list(pool.map(movie, range(thr)))
                t0 = time.perf_counter()
                outs = list(pool.map(movie, range(n)))
                dt = time.perf_counter() - t0
            res.append({"threads": thr, "movies": n, "s_per_movie": round(dt / n, 4),
                        "chain_ms": round(np.mean([o[0] for o in outs]) * 1e3, 1),
                        "fit_ms": round(np.mean([o[1] for o in outs]) * 1e3, 1), "rows": outs[0][2]})
    n_stress = int(os.environ.get("C5_STRESS", "0"))
    for it in range(n_stress):
        thr = 2 + it % 2
        with ThreadPoolExecutor(max_workers=thr) as pool:
            list(pool.map(movie, range(2 * thr)))
for r in res:
    print(json.dumps(r))
import collections
print("(seeds, sweeps, fits, evaluations) per channel of one movie:", INFO[1])
print("rows per channel over %d movies:" % len(ROWS), dict(collections.Counter(ROWS)))
