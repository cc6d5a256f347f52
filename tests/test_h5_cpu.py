"""FOV save-file layer (classes/batch_functions.py:305-556) on the CPU: the HDF5 binding and the helper functions against
fixtures produced by the REFERENCE's functions under h5py (oracle/make_golden_h5.py), plus a live cross-check with the
h5py of /opt/conda when that interpreter exists."""
import os
import pickle
import shutil
import subprocess
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))

from imageanalysis3_amd.io_tools import h5lite as H  # noqa: E402

pytestmark = pytest.mark.skipif(not H.available(), reason="libhdf5 not present")
CONDA_PY = "/opt/conda/bin/python3.9"


from conftest import h5_helper_inputs as helper_inputs  # noqa: E402


def read_group(path, data_type):
    with H.File(path, "r") as f:
        g = f[data_type]
        return {k: g[k][...] for k in g.keys()}


def test_h5lite_reads_reference_written_file():
    """A file made by h5py + the reference's save functions: every dataset, its layout and its growable axis."""
    gold = np.load(os.path.join(GOLD, "h5batch.npz"))
    with H.File(os.path.join(GOLD, "fov_ref.hdf5"), "r") as f:
        assert f.keys() == ['unique'] and 'unique' in f and 'combo' not in f
        g = f['unique']
        assert sorted(g.keys()) == ['channels', 'drifts', 'flags', 'ids', 'ims', 'raw_spots', 'spots']
        for k in ('ids', 'channels', 'spots', 'raw_spots', 'drifts', 'flags', 'ims'):
            got = g[k][...]
            assert got.dtype == gold['h_' + k].dtype and np.array_equal(got, gold['h_' + k]), k
        assert g['spots'].maxshape == (3, None, 11) and g['spots'].chunks is not None
        assert g['ims'].chunks == (1, 4, 8, 8) and g['ims'].dtype == np.uint16
        assert g['channels'].dtype == np.dtype('S3') and g['ids'].dtype == np.int32
        assert np.array_equal(g['ims'][1], gold['h_ims'][1]) and np.array_equal(g['spots'][2, :3, :], gold['h_spots'][2, :3])
        assert g['flags'][0] == gold['h_flags'][0] and len(g['ids']) == 3
        with pytest.raises(KeyError):
            f['combo']
        with pytest.raises(IndexError):
            g['ims'][3]


def test_save_file_helpers_replay_matches_reference(tmp_path):
    """Same call sequence as the reference ran for the fixture, on a file made by create_fov_save_file."""
    from imageanalysis3_amd.classes import batch_functions as B
    gold = np.load(os.path.join(GOLD, "h5batch.npz"))
    path = str(tmp_path / "fov.hdf5")
    B.create_fov_save_file(path, 'unique', [5, 2, 9], ['750', '647', '561'], (4, 8, 8), max_num_seeds=4)
    ims, spots, raw = helper_inputs()
    r1 = B.save_image_to_fov_file(path, ims[:2], 'unique', [2, 5], True, np.array([0.5, -1.5, 2.25]), 0, verbose=False)
    r2 = B.save_image_to_fov_file(path, [ims[2], ims[2]], 'unique', [2, 9], False,
                                  [np.array([9., 9., 9.]), np.array([1., 2., 3.])], 0, verbose=False)
    r3 = B.save_image_to_fov_file(path, [ims[0]], 'unique', [9], False, None, None, verbose=False)
    assert [r1, r2, r3] == list(gold['h_returns'])
    li, lf, ld = B.load_image_from_fov_file(path, 'unique', [9, 5], load_drift=True, verbose=False)
    assert np.array_equal(np.array(li), gold['h_load_ims']) and np.array_equal(np.array(lf), gold['h_load_flags'])
    assert np.array_equal(np.array(ld), gold['h_load_drifts'])
    li1, lf1 = B.load_image_from_fov_file(path, 'unique', 5, verbose=False)          # a bare int id
    assert np.array_equal(li1[0], gold['h_load_ims'][1])
    B.save_spots_to_fov_file(path, spots[:2], 'unique', [5, 9], raw_spot_list=raw[:2], verbose=False)   # grows 4 -> 7
    B.save_spots_to_fov_file(path, [spots[2]], 'unique', [5], raw_spot_list=[raw[2]], verbose=False)    # kept
    B.save_spots_to_fov_file(path, [spots[2]], 'unique', [9], raw_spot_list=[raw[2]], overwrite=True, verbose=False)
    got = read_group(path, 'unique')
    for k in ('ids', 'channels', 'spots', 'raw_spots', 'drifts', 'flags', 'ims'):
        assert got[k].dtype == gold['h_' + k].dtype and np.array_equal(got[k], gold['h_' + k]), k
    # tables whose maxshape forbids growth are recreated (reference :454-475)
    with H.File(path, "a") as f:
        g = f['unique']
        old = g['spots'][...]
        for nm in ('spots', 'raw_spots'):
            del g[nm]
            g.create_dataset(nm, old.shape, dtype='f', maxshape=old.shape, chunks=True)
        g['spots'][...] = old
    big = np.arange(9 * 11, dtype=np.float32).reshape(9, 11) + 1
    B.save_spots_to_fov_file(path, [big], 'unique', [2], raw_spot_list=[big], verbose=False)
    again = read_group(path, 'unique')
    assert again['spots'].shape == (3, 9, 11) and np.array_equal(again['spots'][1], big)
    assert np.array_equal(again['spots'][0, :7], old[0]) and not again['spots'][0, 7:].any()


def test_save_file_helpers_errors(tmp_path):
    from imageanalysis3_amd.classes import batch_functions as B
    path = str(tmp_path / "fov.hdf5")
    im = np.zeros((4, 8, 8), np.uint16)
    with pytest.raises(IOError):
        B.save_image_to_fov_file(path, [im], 'unique', [1], verbose=False)
    with pytest.raises(IOError):
        B.load_image_from_fov_file(path, 'unique', [1], verbose=False)
    with pytest.raises(IOError):
        B.save_spots_to_fov_file(path, [np.zeros((1, 11))], 'unique', [1], verbose=False)
    B.create_fov_save_file(path, 'unique', [1, 2], ['750', '647'], (4, 8, 8), max_num_seeds=4)
    with pytest.raises(ValueError):
        B.save_image_to_fov_file(path, [im], 'nonsense', [1], verbose=False)
    with pytest.raises(ValueError):
        B.save_image_to_fov_file(path, [im, im], 'unique', [1], verbose=False)
    with pytest.raises(IndexError):
        B.save_image_to_fov_file(path, [im, im], 'unique', [1, 2], drift=[np.zeros(3)] * 3, verbose=False)
    with pytest.raises(ValueError):   # id not in the file: list.index
        B.save_image_to_fov_file(path, [im], 'unique', [7], verbose=False)
    with pytest.raises(TypeError):
        B.load_image_from_fov_file(path, 'unique', "1", verbose=False)
    with pytest.raises(KeyError):     # data type allowed but absent from the file
        B.load_image_from_fov_file(path, 'combo', [1], verbose=False)
    with pytest.raises(IndexError):
        B.save_spots_to_fov_file(path, [np.zeros((1, 11))], 'unique', [1], raw_spot_list=[], verbose=False)
    with pytest.raises(ValueError):
        B.create_fov_save_file(path, 'nonsense', [1], ['750'], (4, 8, 8))


def test_drift_file_helpers_match_reference(tmp_path):
    from imageanalysis3_amd.classes import batch_functions as B
    gold = np.load(os.path.join(GOLD, "h5batch.npz"))
    td = str(tmp_path)
    dfile = os.path.join(td, "drift", "drift.pkl")
    assert B.create_drift_file(dfile, os.path.join(td, "H0R0", "Conv_zscan_05.dax"), verbose=False)
    assert B.save_drift_to_file(dfile, os.path.join(td, "H1R1", "Conv_zscan_05.dax"), np.array([1., 2., 3.]), verbose=False)
    assert B.save_drift_to_file(dfile, os.path.join(td, "H1R1", "Conv_zscan_05.dax"), np.array([7., 7., 7.]), verbose=False)
    dd = pickle.load(open(dfile, 'rb'))
    assert sorted(dd.keys()) == list(gold['h_drift_keys'])
    assert np.array_equal(np.array([dd[k] for k in sorted(dd)]), gold['h_drift_vals'])
    B.save_drift_to_file(dfile, os.path.join(td, "H1R1", "Conv_zscan_05.dax"), np.array([7., 7., 7.]), overwrite=True, verbose=False)
    assert np.array_equal(pickle.load(open(dfile, 'rb'))['H1R1/Conv_zscan_05.dax'], [7., 7., 7.])
    B.create_drift_file(dfile, os.path.join(td, "H0R0", "Conv_zscan_05.dax"), verbose=False)   # no update
    assert len(pickle.load(open(dfile, 'rb'))) == 2


def test_h5lite_basics(tmp_path):
    path = str(tmp_path / "a.hdf5")
    with H.File(path, "w") as f:
        g = f.create_group("g")
        with pytest.raises(ValueError):
            f.create_group("g")
        assert f.require_group("g").name == "/g"
        d = g.create_dataset("x", (3, 4), dtype='f8')
        assert d.shape == (3, 4) and d.maxshape == (3, 4) and d.chunks is None and len(d) == 3
        d[...] = np.arange(12).reshape(3, 4)
        d[1, 1:3] = [50, 60]
        d[-1] = 7
        assert np.array_equal(d[:, 1], [1, 50, 7]) and d[1, 2] == 60 and d[()].shape == (3, 4)
        with pytest.raises(TypeError):
            d.resize(5, 0)
        with pytest.raises(NotImplementedError):
            d[::2]
        g['y'] = np.array([1, 2, 3], dtype=np.uint8)
        assert g['y'].dtype == np.uint8 and 'y' in g and 'g/y' in f and 'g/z' not in f
        e = g.create_dataset("e", (0, 11), dtype='f', maxshape=(None, 11))
        assert e.chunks is not None and e[...].shape == (0, 11)
        e.resize((5, 11)); e[4] = 1
        assert e.shape == (5, 11) and e[...].sum() == 11
        f.attrs['name'] = 'fov_05'
        f.attrs['n'] = 3
        g['x'].attrs['scale'] = np.array([200., 108., 108.])
        del g['y']
        with pytest.raises(KeyError):
            g['y']
        with pytest.raises(KeyError):
            del g['y']
    with H.File(path, "r") as f:
        assert f.attrs['name'] == 'fov_05' and f.attrs['n'] == 3 and 'missing' not in f.attrs
        assert np.array_equal(f['g/x'].attrs['scale'], [200., 108., 108.])
        assert np.asarray(f['g']['x']).shape == (3, 4)
        with pytest.raises(OSError):
            f['g']['x'][0] = 1   # read-only file
    with pytest.raises(OSError):
        H.File(str(tmp_path / "missing.hdf5"), "r")
    with pytest.raises(OSError):
        H.File(path, "w-")
    with pytest.raises(ValueError):
        H.File(path, "q")


@pytest.mark.skipif(not os.path.isfile(CONDA_PY), reason="no interpreter with h5py in this environment")
def test_files_interoperate_with_h5py(tmp_path):
    """Written here -> read and extended by h5py -> read back here."""
    from imageanalysis3_amd.classes import batch_functions as B
    path = str(tmp_path / "fov.hdf5")
    B.create_fov_save_file(path, 'unique', [5, 2, 9], ['750', '647', '561'], (4, 8, 8), max_num_seeds=4)
    ims, spots, raw = helper_inputs()
    B.save_image_to_fov_file(path, ims, 'unique', [5, 2, 9], True, np.array([1., 2., 3.]), 0, verbose=False)
    B.save_spots_to_fov_file(path, spots, 'unique', [5, 2, 9], raw_spot_list=raw, verbose=False)
    code = r'''
import sys, zlib, numpy as np, h5py
with h5py.File(sys.argv[1], "a", libver="latest") as f:
    g = f["unique"]
    print(g["ids"][:].tolist(), [c.decode() for c in g["channels"][:]], g["ims"].shape, g["ims"].dtype, g["ims"].chunks,
          g["spots"].shape, g["spots"].maxshape, g["flags"][:].tolist(), g["drifts"][:].tolist(),
          zlib.crc32(g["ims"][:].tobytes()), zlib.crc32(g["spots"][:].tobytes()), zlib.crc32(g["raw_spots"][:].tobytes()))
    g["spots"].resize(12, 1)
    g["spots"][2, 8:12, :] = 3.0
    g["flags"][1] = 1
    g.create_dataset("extra", data=np.arange(5, dtype="i8"))
'''
    out = subprocess.run([CONDA_PY, "-W", "ignore", "-c", code, path], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    import zlib
    g0 = read_group(path, 'unique')
    exp_ims = np.array(ims)
    exp = "%s %s %s %s %s %s %s %s %s" % ([5, 2, 9], ['750', '647', '561'], (3, 4, 8, 8), "uint16", (1, 4, 8, 8), (3, 7, 11),
                                       (3, None, 11), [2, 2, 2], [[1.0, 2.0, 3.0]] * 3)
    assert out.stdout.startswith(exp), out.stdout
    sp = np.zeros((3, 7, 11), np.float32); rw = np.zeros((3, 7, 11), np.float32)
    for i, (s, r) in enumerate(zip(spots, raw)):
        sp[i, :len(s)] = s; rw[i, :len(r)] = r
    assert out.stdout.split()[-3:] == [str(zlib.crc32(exp_ims.tobytes())), str(zlib.crc32(sp.tobytes())), str(zlib.crc32(rw.tobytes()))]
    assert g0['spots'].shape == (3, 12, 11) and (g0['spots'][2, 8:12] == 3.0).all() and np.array_equal(g0['spots'][:, :7], sp)
    assert list(g0['flags']) == [2, 1, 2] and np.array_equal(g0['extra'], np.arange(5))


def test_segmentation_replay_helpers_host_side():
    """Host-side pieces of the segmentation-driven fit replay (tests/harness/replay.py): the label vote around a spot
    (classes/partition_spots.py:113-140 semantics) and the label's bounding box (segmentation_tools/cell.py:598-611)."""
    from harness import replay as R
    lab = np.zeros((6, 20, 20), np.int32)
    lab[:, 2:8, 2:8] = 1
    lab[:, 10:18, 9:16] = 7
    centres = np.array([[3, 4.4, 5.6], [2, 13, 12], [0, 19, 0], [5, 8.6, 8.4]])
    assert list(R.labels_around(lab, centres, 1)) == [1, 7, -1, 7]
    assert R.label_box(lab == 7, 2).tolist() == [[0, 6], [8, 20], [7, 18]]
    assert R.label_box(lab > 0).tolist() == [[0, 6], [1, 19], [1, 17]]
