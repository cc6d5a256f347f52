"""TEST INFRASTRUCTURE ONLY — fixtures for the FOV save-file layer (classes/batch_functions.py:60-556).

The reference's persistence goes through h5py, which the system interpreter of this image lacks; the Anaconda
interpreter has it, so this script runs the REFERENCE's own functions there:

    /opt/conda/bin/python3.9 -W ignore oracle/make_golden_h5.py

and writes tests/golden/h5batch.npz (+ tests/golden/fov_ref.hdf5, a small save file produced by h5py and the reference's
helpers, which the h5lite tests read back).  The save-file group itself is created with the h5py calls of
classes/field_of_view.py:1340-1398 (ids / channels / ims / spots / raw_spots / drifts / flags).
"""
import contextlib
import io
import os
import shutil
import sys
import tempfile
import zlib
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
import ref_loader  # noqa: E402


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def quiet(f, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return f(*a, **k)


def make_save_file(path, data_type, ids, channels, im_size, spot_len):
    """h5py calls of classes/field_of_view.py:1340-1398."""
    import h5py
    n = len(ids)
    with h5py.File(path, "a", libver="latest") as f:
        g = f.create_group(data_type)
        g.create_dataset('ids', (n,), dtype='i', data=ids)
        g.create_dataset('channels', (n,), dtype='S3', data=[str(c).encode('utf8') for c in channels])
        g.create_dataset('ims', (n,) + tuple(im_size), dtype='u2', chunks=(1,) + tuple(im_size))
        g.create_dataset('spots', (n, spot_len, 11), dtype='f', maxshape=(n, None, 11), chunks=True)
        g.create_dataset('raw_spots', (n, spot_len, 11), dtype='f', maxshape=(n, None, 11), chunks=True)
        g.create_dataset('drifts', (n, 3), dtype='f')
        g.create_dataset('flags', (n,), dtype='u1')


def dump(path, data_type, prefix, d, with_ims=True):
    import h5py
    with h5py.File(path, "r") as f:
        g = f[data_type]
        for k in ('ids', 'channels', 'spots', 'raw_spots', 'drifts', 'flags'):
            d[prefix + k] = g[k][:]
        if with_ims:
            d[prefix + 'ims'] = g['ims'][:]
        else:
            d[prefix + 'ims_crc'] = np.array([crc(g['ims'][i]) for i in range(len(g['ims']))], np.uint32)


def helpers_golden(B, d, td):
    path = os.path.join(td, "fov_ref.hdf5")
    make_save_file(path, 'unique', [5, 2, 9], ['750', '647', '561'], (4, 8, 8), 4)
    import conftest as T
    ims, spots, raw = T.h5_helper_inputs()
    r = {}
    r['save1'] = quiet(B.save_image_to_fov_file, path, ims[:2], 'unique', [2, 5], True, np.array([0.5, -1.5, 2.25]), 0)
    # second write without overwrite: slot 2 is kept, slot 9 is new, per-image drifts, unwarped flag
    r['save2'] = quiet(B.save_image_to_fov_file, path, [ims[2], ims[2]], 'unique', [2, 9], False,
                       [np.array([9., 9., 9.]), np.array([1., 2., 3.])], 0)
    r['save3'] = quiet(B.save_image_to_fov_file, path, [ims[0]], 'unique', [9], False, None, None)   # nothing to do
    li, lf, ld = quiet(B.load_image_from_fov_file, path, 'unique', [9, 5], load_drift=True)
    d['h_load_ims'], d['h_load_flags'], d['h_load_drifts'] = np.array(li), np.array(lf), np.array(ld)
    quiet(B.save_spots_to_fov_file, path, spots[:2], 'unique', [5, 9], raw_spot_list=raw[:2])      # 7 > 4: resize
    quiet(B.save_spots_to_fov_file, path, [spots[2]], 'unique', [5], raw_spot_list=[raw[2]])        # kept (non-zero)
    quiet(B.save_spots_to_fov_file, path, [spots[2]], 'unique', [9], raw_spot_list=[raw[2]], overwrite=True)
    d['h_returns'] = np.array([r['save1'], r['save2'], r['save3']])
    dump(path, 'unique', 'h_', d)
    shutil.copy(path, os.path.join(OUT, "fov_ref.hdf5"))
    # drift pickle helpers
    dfile = os.path.join(td, "drift", "drift.pkl")
    quiet(B.create_drift_file, dfile, os.path.join(td, "H0R0", "Conv_zscan_05.dax"))
    quiet(B.save_drift_to_file, dfile, os.path.join(td, "H1R1", "Conv_zscan_05.dax"), np.array([1., 2., 3.]))
    quiet(B.save_drift_to_file, dfile, os.path.join(td, "H1R1", "Conv_zscan_05.dax"), np.array([7., 7., 7.]))   # kept
    import pickle
    dd = pickle.load(open(dfile, 'rb'))
    d['h_drift_keys'] = np.array(sorted(dd.keys()))
    d['h_drift_vals'] = np.array([dd[k] for k in sorted(dd.keys())])


def batch_golden(B, d, td):
    import conftest as T
    import h5py
    case, size, corr, corr_nowarp, fit = T.batch_inputs()
    os.makedirs(os.path.join(td, "H1R1"))
    movie = os.path.join(td, "H1R1", "Conv_zscan_05.dax")
    T.write_dax(movie, case["raw"])
    ref_im = np.zeros(size, np.uint16)   # reference bead image: unused, every variant has a stored drift
    # verbose=True for the warped variant: the reference only warps inside `if verbose:` (io_tools/load.py:438-453)
    for tag, warp, cargs in (("w_", True, corr), ("n_", False, corr_nowarp)):
        path = os.path.join(td, tag + "fov.hdf5")
        make_save_file(path, 'unique', [5, 2, 9], ['750', '647', '561'], size, 4)
        with h5py.File(path, "a", libver="latest") as f:   # a drift stored by an earlier pass
            f['unique']['drifts'][:2, :] = np.array(case["drift"], np.float32)
        quiet(B.batch_process_image_to_spots, movie, ['750', '647'], path, 'unique', [5, 2], ref_im,
              warp_image=warp, correction_args=dict(cargs), fitting_args=dict(fit), verbose=warp)
        dump(path, 'unique', tag, d, with_ims=False)
        # second pass over the same movie: images are carried over from the file, spot tables are kept
        quiet(B.batch_process_image_to_spots, movie, ['750', '647'], path, 'unique', [5, 2], ref_im,
              warp_image=warp, correction_args=dict(cargs), fitting_args=dict(fit, max_num_seeds=3), verbose=warp)
        dump(path, 'unique', tag + "again_", d, with_ims=False)
        # third: overwrite the spots with the shorter tables
        quiet(B.batch_process_image_to_spots, movie, ['750', '647'], path, 'unique', [5, 2], ref_im,
              warp_image=warp, correction_args=dict(cargs), fitting_args=dict(fit, max_num_seeds=3),
              overwrite_spot=True, verbose=warp)
        dump(path, 'unique', tag + "over_", d, with_ims=False)
    # drift measured by the reference itself (phase correlation of the bead channel, scikit-image 0.18.3) instead of a
    # stored one: nothing is given, everything downstream (warp, fit, stored drifts) hangs on that measurement
    nb, Z = case["nb"], case["Z"]
    bead = np.ascontiguousarray(case["raw"][nb + (3 - nb) % 4::4][:Z])
    bead_ref = np.roll(bead, (1, -2), axis=(1, 2))
    path = os.path.join(td, "d_fov.hdf5")
    make_save_file(path, 'unique', [5, 2, 9], ['750', '647', '561'], size, 4)
    quiet(B.batch_process_image_to_spots, movie, ['750', '647'], path, 'unique', [5, 2], bead_ref,
          warp_image=True, correction_args=dict(corr), fitting_args=dict(fit), verbose=True)
    dump(path, 'unique', "d_", d, with_ims=False)
    d['raw_crc'] = crc(case["raw"])


def phase_golden():
    """skimage.registration.phase_cross_correlation (the real one: 0.18.3 in /opt/conda, the un-normalised
    correlation of the 0.17/0.18 releases) and the reference's align_image on its default phase-correlation path
    (correction_tools/alignment.py:527-695), on the bead pairs of imageanalysis3_amd.synth.make_bead_pair."""
    import skimage
    from skimage.registration import phase_cross_correlation
    from imageanalysis3_amd import synth
    R = ref_loader.load_reference()
    d = {"skimage_version": np.array(skimage.__version__)}
    dd = np.array([0.7, -3.25, 5.5])
    ref, src, _, _ = synth.make_bead_pair((20, 96, 96), 20, 3, dd, margin=(5, 12, 12), min_sep=12.0)
    for tag, a, b in (("f32", ref, src), ("u16", ref.astype(np.uint16), src.astype(np.uint16))):
        for up in (1, 10, 100):
            sh, err, ph = phase_cross_correlation(a, b, upsample_factor=up)
            d["pcc_%s_%d" % (tag, up)] = np.concatenate([np.asarray(sh, np.float64), [err, ph]])
    d2 = np.array([1.3, -4.6, 7.25])
    ref2, src2, _, _ = synth.make_bead_pair((30, 256, 256), 120, 21, d2)
    crops = R.alignment.generate_drift_crops(single_im_size=[30, 256, 256])
    d["align_crops"] = np.asarray(crops)
    out = quiet(R.alignment.align_image, src2, ref2, crop_list=crops, use_autocorr=True, verbose=False)
    d["align_drift"], d["align_flag"] = np.asarray(out[0], np.float64), np.array(out[1])
    out = quiet(R.alignment.align_image, src2.astype(np.uint16), ref2.astype(np.uint16), crop_list=crops, use_autocorr=True,
                verbose=False)
    d["align_u16_drift"], d["align_u16_flag"] = np.asarray(out[0], np.float64), np.array(out[1])
    np.savez_compressed(os.path.join(OUT, "phase.npz"), **d)
    for k in sorted(d):
        print(k, d[k])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "phase":
        phase_golden()
        sys.exit(0)
    B = ref_loader.load_batch()
    d = {}
    with tempfile.TemporaryDirectory() as td:
        helpers_golden(B, d, td)
        batch_golden(B, d, td)
    np.savez_compressed(os.path.join(OUT, "h5batch.npz"), **d)
    for k in sorted(d):
        print(k, np.shape(d[k]), getattr(d[k], 'dtype', None))
