"""Stress of the per-thread streams + stream-ordered cache (developer tool): many small movies through
batch_process_images_to_spots with 1 and 8 threads, repeated; file contents must be identical."""
import contextlib, io, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import conftest as T
from imageanalysis3_amd.classes import batch_functions as B
from imageanalysis3_amd.io_tools import h5lite as H
case, size, corr, corr_nowarp, fit = T.batch_inputs()
nb, Z = case["nb"], case["Z"]
bead = np.ascontiguousarray(case["raw"][nb + (3 - nb) % 4::4][:Z])
bead_ref = np.roll(bead, (1, -2), axis=(1, 2))
N = 32
with tempfile.TemporaryDirectory() as td:
    movies, ids = [], []
    rng = np.random.RandomState(0)
    for r in range(N):
        os.makedirs(os.path.join(td, "H%dR%d" % (r, r)))
        m = os.path.join(td, "H%dR%d" % (r, r), "Conv_zscan_05.dax")
        raw = case["raw"].copy()
        raw[nb:-nb] = np.roll(raw[nb:-nb], int(rng.randint(0, 5)) * 4, axis=0)   # a different movie per round
        T.write_dax(m, raw)
        movies.append(m); ids.append([10 + 2 * r, 11 + 2 * r])
    all_ids = [i for p in ids for i in p]

    def run(path, threads):
        B.create_fov_save_file(path, 'unique', all_ids, ['750', '647'] * N, size, max_num_seeds=4)
        args = [dict(dax_filename=m, sel_channels=['750', '647'], region_ids=i) for m, i in zip(movies, ids)]
        shared = dict(save_filename=path, data_type='unique', ref_filename=bead_ref, warp_image=True,
                      correction_args=dict(corr), fitting_args=dict(fit), verbose=True)
        with contextlib.redirect_stdout(io.StringIO()):
            B.batch_process_images_to_spots(args, num_threads=threads, shared_kwargs=shared)
        with H.File(path, "r") as f:
            return {k: f['unique'][k][...] for k in ('ims', 'spots', 'raw_spots', 'flags', 'drifts')}
    ref = run(os.path.join(td, "seq.hdf5"), 1)
    for rep in range(3):
        got = run(os.path.join(td, "par%d.hdf5" % rep), 8)
        bad = [k for k in ref if not np.array_equal(ref[k], got[k])]
        print("repeat", rep, "mismatching datasets:", bad)
        assert not bad
print("stress ok")
