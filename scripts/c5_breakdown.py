"""Where one C5 movie spends its time (developer tool): stage timers of libia3 + wall clock of the Python steps."""
import contextlib, io, json, os, sys, tempfile, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("C5_MOVIES", "2")
from imageanalysis3_amd import synth, _lib as L
from imageanalysis3_amd.io_tools.load import DeviceBuffer
from imageanalysis3_amd.classes import batch_functions as B
Z, X, Y, NB = 50, 2048, 2048, 10
chs = ['750', '647', '561', '488']
L.check(L.lib().ia3_init(0))
ims = [synth.make_fov((Z, X, Y), 5000 if i < 3 else 300, 40 + i, dtype=np.uint16)[0] for i in range(4)]
raw = np.zeros((NB + 4 * Z + NB, X, Y), np.uint16)
for i in range(4):
    start = NB + (i - NB) % 4
    raw[start:start + 4 * Z:4] = ims[i]
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
with tempfile.TemporaryDirectory() as td:
    os.makedirs(os.path.join(td, "H1R1"))
    movie = os.path.join(td, "H1R1", "Conv_zscan_05.dax")
    raw.tofile(movie)
    with open(movie[:-4] + ".inf", "w") as f:
        f.write("frame dimensions = %d x %d\nnumber of frames = %d\n" % (Y, X, raw.shape[0]))
    corr = dict(single_im_size=[Z, X, Y], all_channels=chs, num_buffer_frames=NB, num_empty_frames=0,
                corr_channels=chs[:3], illumination_profile=illum, bleed_profile=bleed, chromatic_profile=chrom)
    path = os.path.join(td, "fov.hdf5")
    for rep in range(2):
        B.create_fov_save_file(path, 'unique', [0, 1, 2], chs[:3], [Z, X, Y], max_num_seeds=6000, overwrite=True)
        L.profile_enable(True); L.profile_collect()
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            pr.enable()
            B.batch_process_image_to_spots(movie, chs[:3], path, 'unique', [0, 1, 2], ims[3], warp_image=True, save_image=False,
                                           correction_args=dict(corr), fitting_args=dict(max_num_seeds=None, seeding_kwargs={}), verbose=True)
            pr.disable()
        dt = time.perf_counter() - t0
        prof = L.profile_collect(); L.profile_enable(False)
    print("wall %.3f s; device stage totals (ms):" % dt, {k: round(v[1], 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:14]}, "sum %.1f" % sum(v[1] for v in prof.values()))
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[-3500:])
