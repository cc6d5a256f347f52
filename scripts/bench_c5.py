"""BASELINE.json configs[4] on one GPU (developer benchmark, not the driver's bench.py): one round-folder movie =
4-colour uint16 .dax (3 signal channels + beads) -> hot pixels, bleedthrough, illumination, bead drift (phase
correlation against the reference round), cubic warp with drift + dense chromatic field, DoG seed + LM fit, images +
drift + spots into the FOV save file, through `batch_process_image_to_spots`.  Prints one JSON line per variant."""
import json, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis3_amd import synth, _lib as L
from imageanalysis3_amd.io_tools.load import DeviceBuffer
from imageanalysis3_amd.classes import batch_functions as B

Z = int(os.environ.get("C5_Z", "50")); X = Y = int(os.environ.get("C5_XY", "2048")); NB = 10
N_MOVIES = int(os.environ.get("C5_MOVIES", "8"))
chs = ['750', '647', '561', '488']
L.check(L.lib().ia3_init(0))
t0 = time.time()
ims = [synth.make_fov((Z, X, Y), 5000 if i < 3 else 300, 40 + i, dtype=np.uint16)[0] for i in range(4)]
raw = np.zeros((NB + 4 * Z + NB, X, Y), np.uint16)
for i in range(4):
    start = NB + (i - NB) % 4
    raw[start:start + 4 * Z:4] = ims[i]
rng = np.random.RandomState(0)
yy, xx = np.meshgrid(np.linspace(-1, 1, Y), np.linspace(-1, 1, X))
bump = (0.55 + 0.45 * np.exp(-(xx ** 2 + yy ** 2))).astype(np.float32)
illum = {c: DeviceBuffer(bump) for c in chs}
bleed = np.zeros((3, 3, X, Y), np.float32)
for a in range(3):
    for b in range(3):
        bleed[a, b] = 1.0 if a == b else 0.05
bleed = DeviceBuffer(bleed)
zz = np.linspace(-1, 1, Z, dtype=np.float32)[:, None, None]
field = np.stack([0.2 * zz + 0 * xx[None].astype(np.float32), (0.8 * xx[None] + 0 * zz).astype(np.float32),
                  (0.8 * yy[None] + 0 * zz).astype(np.float32)]).astype(np.float32)
chrom = {'750': DeviceBuffer(field), '647': None, '561': DeviceBuffer(-field)}
ref_bead = ims[3]
print("synthesis %.1f s" % (time.time() - t0), flush=True)

with tempfile.TemporaryDirectory(dir=os.environ.get("C5_TMP", "/tmp")) as td:
    movie0 = os.path.join(td, "movie.dax")
    raw.tofile(movie0)
    with open(movie0[:-4] + ".inf", "w") as f:
        f.write("frame dimensions = %d x %d\nnumber of frames = %d\n" % (Y, X, raw.shape[0]))
    movies = []
    for r in range(N_MOVIES):
        d = os.path.join(td, "H%dR%d" % (r, r)); os.makedirs(d)
        m = os.path.join(d, "Conv_zscan_05.dax")
        os.symlink(movie0, m); os.symlink(movie0[:-4] + ".inf", m[:-4] + ".inf")
        movies.append(m)
    del raw
    corr = dict(single_im_size=[Z, X, Y], all_channels=chs, num_buffer_frames=NB, num_empty_frames=0,
                corr_channels=chs[:3], illumination_profile=illum, bleed_profile=bleed, chromatic_profile=chrom)
    fit = dict(max_num_seeds=None, seeding_kwargs={})

    def run(tag, threads, save_image, n):
        path = os.path.join(td, tag + ".hdf5")
        ids = list(range(3 * n))
        B.create_fov_save_file(path, 'unique', ids, chs[:3] * n, [Z, X, Y], max_num_seeds=6000, overwrite=True)
        args = [dict(dax_filename=movies[r], sel_channels=chs[:3], region_ids=ids[3 * r:3 * r + 3]) for r in range(n)]
        shared = dict(save_filename=path, data_type='unique', ref_filename=ref_bead, warp_image=True, save_image=save_image,
                      correction_args=dict(corr), fitting_args=dict(fit), verbose=True)
        import contextlib, io
        L.check(L.lib().ia3_sync()); t = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            B.batch_process_images_to_spots(args, num_threads=threads, shared_kwargs=shared)
        dt = time.perf_counter() - t
        from imageanalysis3_amd.io_tools import h5lite as H
        with H.File(path, "r") as f:
            sp = f['unique']['spots'][...]
            drifts = f['unique']['drifts'][...]
        rows = int(sp.any(axis=2).sum())
        print(json.dumps({"variant": tag, "movies": n, "threads": threads, "save_image": save_image,
                          "s_per_movie": round(dt / n, 3), "images_per_s": round(3 * n / dt, 2), "spots_per_s": round(rows / dt, 1),
                          "rows": rows, "max_abs_drift": float(np.abs(drifts).max())}), flush=True)
        os.remove(path)

    run("warmup", 1, False, 1)   # every variant must report the same number of rows per movie
    run("seq_nosave", 1, False, N_MOVIES)
    run("thr4_nosave", 4, False, N_MOVIES)
    run("thr8_nosave", 8, False, N_MOVIES)
    run("thr4_save", 4, True, N_MOVIES)
