"""Per-movie production entry and FOV save-file helpers (interface of the reference's classes/batch_functions.py).

``batch_process_image_to_spots`` (reference :60-303) is what ``Field_of_View._process_image_to_spots`` hands to its
process pool for every (round folder, FOV) movie: see what the FOV save file already holds, correct the channels that
are still missing, store images / drift / warp flags, fit every channel and store ``spots`` / ``raw_spots``.  Public
names, argument order, exceptions and every rule that decides what is (re)written follow the reference; the bodies
are organised around this package instead: a ``SaveFile`` helper owns the HDF5 bookkeeping (``io_tools.h5lite``, libhdf5
through ctypes — there is no h5py for the system interpreter), corrected channels stay resident on the GPU between
``correct_fov_image(..., return_device=True)`` and ``fit_fov_image``, and ``batch_process_images_to_spots`` replaces the
process pool by threads with their own HIP streams.

``create_fov_save_file`` lays out the data-type group the way ``Field_of_View._save_to_file`` does
(classes/field_of_view.py:1314-1398), so a save file can be made without that (out-of-scope) class.
"""
import contextlib
import os
import pickle
import time
import numpy as np
from scipy import ndimage

from . import _allowed_kwds, _max_num_seeds
from .. import _image_dtype
from .. import _lib as L
from ..io_tools import h5lite
from ..io_tools.load import correct_fov_image, _dax_info
from ..spot_tools.fitting import fit_fov_image, get_centers  # noqa: F401  (importable from here, as in the reference)

# seeding threshold per channel (reference :10-17)
Channel_2_SeedTh = {'750': 600, '647': 600, '561': 600, '748': 1000, '637': 1000, '545': 1000}

FLAG_EMPTY, FLAG_UNWARPED, FLAG_WARPED = 0, 1, 2   # `flags` dataset: what the `ims` slot holds


# ----------------------------------------------------------------------------------------------------------------------
# save file
# ----------------------------------------------------------------------------------------------------------------------
def _require_file(filename, what):
    if not os.path.isfile(filename):
        raise IOError(f"{what} file: {filename} doesn't exist!")


def _require_type(data_type):
    if data_type not in _allowed_kwds:
        raise ValueError(f"Wrong input data_type:{data_type}, should be among {_allowed_kwds}.")


class SaveFile(object):
    """One data-type group of a FOV save file, opened for update.  ``row(id)`` maps a region id to its slot
    (``ValueError`` for an id the file does not list, as ``list.index`` in the reference)."""

    def __init__(self, filename, data_type, create_group=False):
        self._file = h5lite.File(filename, "a", libver='latest')
        try:
            self.group = self._file.require_group(data_type) if create_group else self._file[data_type]
        except Exception:
            self._file.close()
            raise
        self._ids = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self._file.close()
        return False

    def row(self, region_id):
        if self._ids is None:
            self._ids = [int(v) for v in self.group['ids'][...]]
        return self._ids.index(region_id)

    def __getitem__(self, name):
        return self.group[name]

    def has(self, name):
        return name in self.group


def create_fov_save_file(filename, data_type, ids, channels, single_im_size, max_num_seeds=None, overwrite=False):
    """Create (or complete) the ``data_type`` group: ``ids (n,) i4``, ``channels (n,) S3``, ``ims (n,Z,X,Y) u2`` with
    one chunk per image, ``spots`` / ``raw_spots`` ``(n, L, 11) f4`` growable along axis 1 (L = ``max_num_seeds``,
    default ``classes._max_num_seeds``), ``drifts (n,3) f4``, ``flags (n,) u1`` (0 empty, 1 unwarped, 2 warped image).
    Members that exist are left alone."""
    _require_type(data_type)
    if len(ids) != len(channels):
        raise ValueError("ids and channels should have the same length")
    n = len(ids)
    zxy = tuple(int(v) for v in single_im_size)
    table_len = int(_max_num_seeds if max_num_seeds is None else max_num_seeds)
    layout = [
        ('ids', dict(shape=(n,), dtype='i', data=np.asarray(ids, dtype=np.int32))),
        ('channels', dict(shape=(n,), dtype='S3', data=[str(c).encode('utf8') for c in channels])),
        ('ims', dict(shape=(n,) + zxy, dtype='u2', chunks=(1,) + zxy)),
        ('spots', dict(shape=(n, table_len, 11), dtype='f', maxshape=(n, None, 11), chunks=True)),
        ('raw_spots', dict(shape=(n, table_len, 11), dtype='f', maxshape=(n, None, 11), chunks=True)),
        ('drifts', dict(shape=(n, 3), dtype='f')),
        ('flags', dict(shape=(n,), dtype='u1')),
    ]
    with h5lite.File(filename, "w" if overwrite else "a", libver='latest') as f:
        grp = f.require_group(data_type)
        for name, spec in layout:
            if name not in grp:
                grp.create_dataset(name, **spec)
    return filename


def save_image_to_fov_file(filename, ims, data_type, region_ids,
                           warp_image=False, drift=None, drift_flag=None,
                           overwrite=False, verbose=True):
    """reference :305-368.  Images go into the slots of ``region_ids`` that are still empty (flag 0) or into all of
    them with ``overwrite``; the slot's flag becomes 2 (warped) or 1; ``drift`` (one vector, or one per image) is stored
    next to every image written.  Returns True when something was written."""
    _require_file(filename, "save")
    _require_type(data_type)
    if len(ims) != len(region_ids):
        raise ValueError(f"Wrong input region_ids:{region_ids}, should of same length as ims, len={len(ims)}.")
    per_image_drift = None
    if drift is not None:
        if np.ndim(drift) == 1:
            per_image_drift = [drift] * len(ims)
        elif len(drift) == len(ims):
            per_image_drift = drift
        else:
            raise IndexError("Length of drift should match ims")
    t0 = time.time()
    if verbose:
        print(f"- writting {data_type} info to file:{filename}")
    written, with_drift = [], []
    new_flag = FLAG_WARPED if warp_image else FLAG_UNWARPED
    with SaveFile(filename, data_type, create_group=True) as sf:
        for k, (rid, im) in enumerate(zip(region_ids, ims)):
            slot = sf.row(rid)
            if not overwrite and sf['flags'][slot] != FLAG_EMPTY:
                continue
            sf['ims'][slot] = im
            sf['flags'][slot] = new_flag
            written.append(rid)
            if per_image_drift is not None:
                sf['drifts'][slot, :] = per_image_drift[k]
                with_drift.append(rid)
    if verbose:
        if written:
            print(f"-- updated ims for id:{written}, drifts for id:{with_drift} in {time.time()-t0:.3f}s")
        else:
            print("-- images and drifts already exist, skip.")
    return len(written) > 0


def load_image_from_fov_file(filename, data_type, region_ids,
                             image_dtype=_image_dtype, load_drift=False, verbose=True):
    """reference :371-419.  ``(images, flags)`` (and ``drifts`` with ``load_drift``) of the given region ids, in the
    order given; ``region_ids``: an int, a list or an array."""
    _require_file(filename, "load")
    _require_type(data_type)
    if isinstance(region_ids, (int, np.integer)):
        wanted = [int(region_ids)]
    elif isinstance(region_ids, (list, np.ndarray)):
        wanted = [int(v) for v in region_ids]
    else:
        raise TypeError(f"Wrong input type for region_ids:{region_ids}")
    t0 = time.time()
    if verbose:
        print(f"- loading {data_type} info from file:{os.path.basename(filename)}", end=' ')
    images, flags, drifts = [], [], []
    with SaveFile(filename, data_type) as sf:
        for rid in wanted:
            slot = sf.row(rid)
            images.append(sf['ims'][slot])
            flags.append(sf['flags'][slot])
            if load_drift:
                drifts.append(sf['drifts'][slot, :])
    if verbose:
        print(f"in {time.time()-t0:.3f}s.")
    return (images, flags, drifts) if load_drift else (images, flags)


def _grow_spot_tables(sf, data_type, need, verbose):
    """Make ``spots`` / ``raw_spots`` at least ``need`` rows long (reference :448-477): resize when the dataset may
    grow, otherwise replace both by growable copies."""
    n, have, width = sf['spots'].shape
    if have >= need:
        return
    limit = sf['spots'].maxshape[1]
    if limit is not None and limit < need:
        if verbose:
            print(f"-- recreate {data_type}_spots and {data_type}_raw_spots from {have} to {need}.")
        for name in ('spots', 'raw_spots'):
            old = sf[name][...]
            del sf.group[name]
            sf.group.create_dataset(name, (n, need, width), dtype='f', maxshape=(n, None, width), chunks=True)
            sf[name][:, :have, :] = old
    else:
        if verbose:
            print(f"-- resize {data_type}_spots and {data_type}_raw_spots from {have} to {need}.")
        for name in ('spots', 'raw_spots'):
            sf[name].resize(need, 1)


def save_spots_to_fov_file(filename, spot_list, data_type, region_ids,
                           raw_spot_list=None,
                           overwrite=False, verbose=True):
    """reference :422-493.  A spot table is written into its region's row when that row is still all zero (or with
    ``overwrite``); rows keep whatever lies beyond the new table's length; both tables grow when a list is longer than
    what is stored.  ``raw_spots`` (coordinates before the chromatic / drift translation) follows the same rule."""
    _require_file(filename, "save")
    _require_type(data_type)
    if len(spot_list) != len(region_ids):
        raise ValueError(f"Wrong input region_ids:{region_ids}, should of same length as spots, len={len(spot_list)}.")
    if raw_spot_list is not None and len(raw_spot_list) != len(spot_list):
        raise IndexError(f"length of input spot_list and raw_spot list should match, {len(spot_list)}, {len(raw_spot_list)}")
    t0 = time.time()
    if verbose:
        print(f"- writting {data_type} spots into file:{filename}")
    updated = []
    with SaveFile(filename, data_type) as sf:
        for k, (rid, table) in enumerate(zip(region_ids, spot_list)):
            _grow_spot_tables(sf, data_type, len(table), verbose)
            slot = sf.row(rid)
            if overwrite or not np.sum(sf['spots'][slot]):
                sf['spots'][slot, :len(table), :] = table
                updated.append(rid)
            if sf.has('raw_spots') and (overwrite or not np.sum(sf['raw_spots'][slot])):
                raw = raw_spot_list[k]          # TypeError when no raw list was given, as in the reference
                sf['raw_spots'][slot, :len(raw), :] = raw
    if verbose:
        print(f"-- updated spots for id:{updated} in {time.time()-t0:.3f}s")
    return True


# ----------------------------------------------------------------------------------------------------------------------
# drift file: pickled {'<round folder>/<movie>.dax': drift}
# ----------------------------------------------------------------------------------------------------------------------
def _drift_key(image_filename):
    folder, name = os.path.split(image_filename)
    return os.path.join(os.path.basename(folder), name)


def _read_drift_dict(drift_filename):
    if not os.path.isfile(drift_filename):
        return {}
    with open(drift_filename, 'rb') as fh:
        return pickle.load(fh)


def _write_drift_dict(drift_filename, table):
    with open(drift_filename, 'wb') as fh:
        pickle.dump(table, fh)


def save_drift_to_file(drift_filename, image_filename, drift, overwrite=False, verbose=True):
    """reference :496-519 — record the drift of one movie unless it is already there."""
    table = _read_drift_dict(drift_filename)
    key = _drift_key(image_filename)
    if overwrite or key not in table:
        table[key] = drift
        if verbose:
            print(f"-- update drift of {key} into file:{drift_filename}")
        _write_drift_dict(drift_filename, table)
    elif verbose:
        print("-- no updates in drift, skip.")
    return True


def create_drift_file(drift_filename, ref_filename,
                      n_dim=3,
                      overwrite=False, verbose=True):
    """reference :523-556 — start (or restart with ``overwrite``) the drift table with a zero drift for the
    reference movie; creates the folder when needed."""
    table = {} if overwrite else _read_drift_dict(drift_filename)
    key = _drift_key(ref_filename)
    if key in table:
        if verbose:
            print(f"-- no updates in drift file:{drift_filename}, skip.")
        return True
    table[key] = np.zeros(n_dim)
    folder = os.path.dirname(drift_filename)
    if not os.path.isdir(folder):
        if verbose:
            print(f"--- creating folder:{folder}")
        os.makedirs(folder)
    if verbose:
        print(f"-- create drift file:{drift_filename} with reference:{key}")
    _write_drift_dict(drift_filename, table)
    return True


# ----------------------------------------------------------------------------------------------------------------------
# one movie
# ----------------------------------------------------------------------------------------------------------------------
def _check_batch_arguments(dax_filename, save_filename, ref_filename, sel_channels, region_ids):
    """The reference's argument checks (:92-118), same exception types and texts."""
    if not os.path.isfile(dax_filename):
        raise IOError(f"Dax file: {dax_filename} is not a file, exit!")
    if not isinstance(dax_filename, str) or not dax_filename.endswith('.dax'):
        raise IOError(f"Dax file: {dax_filename} has wrong data type, exit!")
    if not os.path.isfile(save_filename):
        raise IOError(f"HDF5 file: {save_filename} is not a file, exit!")
    if not isinstance(save_filename, str) or not save_filename.endswith('.hdf5'):
        raise IOError(f"HDF5 file: {save_filename} has wrong data type, exit!")
    if isinstance(ref_filename, str):
        if not os.path.isfile(ref_filename):
            raise IOError(f"Dax file: {ref_filename} is not a file, exit!")
        if not ref_filename.endswith('.dax'):
            raise IOError(f"Dax file: {ref_filename} has wrong data type, exit!")
    elif not isinstance(ref_filename, np.ndarray):
        raise TypeError(f"ref_filename should be np.ndarray or string of path, but {type(ref_filename)} is given")
    if len(region_ids) != len(sel_channels):
        raise ValueError(f"Wrong input region_ids:{region_ids}, should of same length as sel_channels:{sel_channels}.")


@contextlib.contextmanager
def _held(lock):
    if lock is None:
        yield
    else:
        lock.acquire()
        try:
            yield
        finally:
            lock.release()


def _stored_drift(drifts, overwrite_drift):
    """(drift to start from, measure it?) from the drifts stored for this movie's regions (:153-166): one common
    non-zero vector is trusted, anything else means the drift has to be measured."""
    distinct = list(set(tuple(d) for d in drifts))
    if len(distinct) == 1 and np.any(distinct[0]) and not overwrite_drift:
        return np.array(distinct[0]), False
    return np.zeros(len(distinct[0])), True


def batch_process_image_to_spots(dax_filename,
                                 sel_channels,
                                 save_filename,
                                 data_type,
                                 region_ids,
                                 ref_filename,
                                 load_file_lock=None,
                                 warp_image=True,
                                 correction_args={},
                                 save_image=True,
                                 empty_value=0,
                                 fov_savefile_lock=None,
                                 overwrite_image=False,
                                 drift_args={},
                                 save_drift=True,
                                 drift_filename=None,
                                 drift_file_lock=None,
                                 overwrite_drift=False,
                                 fit_spots=True,
                                 fit_in_mask=False,
                                 fitting_args={},
                                 save_spots=True,
                                 spot_file_lock=None,
                                 overwrite_spot=False,
                                 verbose=False,
                                 return_spots=False,
                                 fit_workers=None):
    """reference :60-303 — one movie: corrected images + drift into the FOV save file, then the spots of every selected
    channel.  Returns None like the reference; ``return_spots=True`` (an extension) returns ``(spots, raw_spots)``;
    ``fit_workers`` (an extension): host threads fitting the channels of this movie side by side (default: one per
    channel, at most 4; 1 = one after the other)."""
    _check_batch_arguments(dax_filename, save_filename, ref_filename, sel_channels, region_ids)
    channels = [str(c) for c in sel_channels]
    region_ids = [int(r) for r in region_ids]
    if verbose:
        print(f"+ batch process image: {dax_filename} for channels:{channels}")

    # ---- what does the save file already hold? (:121-166) ----------------------------------------------------------
    # The reference loads every image slot and then decides; a slot is reusable only when it holds data AND its flag
    # says it was stored the way this call wants it (warped or not), so slots whose flag does not match — every slot
    # of a fresh file — are not read at all (3 x 419 MB per movie for a full-size FOV).
    _require_type(data_type)
    redo_everything = overwrite_image or overwrite_drift
    todo, reused, stored_drifts = [], {}, []   # per channel: True = correct it from the movie, False = reuse the stored image
    t_load = time.time()
    with _held(fov_savefile_lock), SaveFile(save_filename, data_type) as sf:
        for c, rid in zip(channels, region_ids):
            slot = sf.row(rid)
            stored_drifts.append(sf['drifts'][slot, :])
            usable = False
            if not redo_everything and int(sf['flags'][slot]) - 1 == int(warp_image):
                im = sf['ims'][slot]
                usable = bool(im.any() if empty_value == 0 else (im != empty_value).any())
                if usable:
                    reused[c] = im
            todo.append(not usable)
    if verbose:
        print(f"- loading {data_type} info from file:{os.path.basename(save_filename)} in {time.time()-t_load:.3f}s.")
    start_drift, measure_drift = _stored_drift(stored_drifts, overwrite_drift)

    resident = []        # DeviceStacks owned by this call
    try:
        # ---- correct the missing channels; they stay on the device for the fit (:169-206) ---------------------------
        fresh, fresh_funcs = {}, {}
        drift, drift_flag = np.array(start_drift), 0
        new_channels = [c for c, t in zip(channels, todo) if t]
        if new_channels:
            if verbose:
                print(f"-- {new_channels} images are required to process, {list(reused)} images are loaded from save file: {save_filename}")
            out = correct_fov_image(dax_filename, new_channels, load_file_lock=load_file_lock,
                                    calculate_drift=measure_drift, drift=start_drift, ref_filename=ref_filename,
                                    warp_image=warp_image, return_drift=True, verbose=verbose, return_device=True,
                                    **correction_args, **drift_args)
            if warp_image:
                stacks, drift, drift_flag = out
            else:
                stacks, funcs, drift, drift_flag = out
                fresh_funcs = dict(zip(new_channels, funcs))
            resident = list(stacks)
            fresh = dict(zip(new_channels, stacks))
        images = [fresh[c] if t else reused[c] for c, t in zip(channels, todo)]   # DeviceStack or host array

        translate = None
        if not warp_image:   # spots of unwarped images are moved instead: one function per channel (:219-229)
            from ..correction_tools.chromatic import generate_chromatic_function
            translate = [fresh_funcs[c] if t else
                         generate_chromatic_function(correction_args['chromatic_profile'][str(c)], drift)
                         for c, t in zip(channels, todo)]

        # ---- images + drift into the save file (:232-245) ----------------------------------------------------------
        if save_image:
            host = [im.download() if isinstance(im, L.DeviceStack) else im for im in images]
            with _held(fov_savefile_lock):
                save_image_to_fov_file(save_filename, host, data_type, region_ids,
                                       warp_image, drift, drift_flag, overwrite_image, verbose)
            del host

        # ---- fit (:248-300) --------------------------------------------------------------------------------------
        spot_list, raw_spot_list = np.array([]), []
        if fit_spots:
            if fit_in_mask:
                if fitting_args.get('seed_mask', None) is None:
                    raise KeyError("seed_mask should be given if fit_in_mask specified")
                if not warp_image:   # the mask was drawn on registered images: move it onto this unwarped one
                    if verbose:
                        print(f"-- start traslating seed_mask by drift: {drift}", end=' ')
                        _translate_start = time.time()
                    fitting_args['seed_mask'] = ndimage.shift(fitting_args['seed_mask'], -drift, mode='constant', cval=0)
                if verbose:   # (the reference reads the timer here on both branches: NameError with warp_image)
                    print(f"-- in {time.time()-_translate_start:.2f}s.")
            def fit_channel(k):
                # the reference sets fitting_args['th_seed'] per channel inside its loop; every task gets its own copy
                kw = dict(fitting_args, th_seed=Channel_2_SeedTh[str(channels[k])])
                return fit_fov_image(images[k], channels[k], verbose=verbose, **kw)

            workers = min(len(images), 4 if fit_workers is None else int(fit_workers))
            if workers > 1:   # the channels of one movie are independent: their fits share the device (own streams)
                from concurrent.futures import ThreadPoolExecutor
                L.check(L.lib().ia3_sync())   # the corrected stacks were produced on THIS thread's stream: finish them
                with ThreadPoolExecutor(max_workers=workers) as pool:
                    raw_spot_list = list(pool.map(fit_channel, range(len(images))))
            else:
                raw_spot_list = [fit_channel(k) for k in range(len(images))]
            if channels:
                fitting_args['th_seed'] = Channel_2_SeedTh[str(channels[-1])]   # what the reference's loop leaves behind
            spot_list = [raw.copy() if translate is None else translate[k](raw) for k, raw in enumerate(raw_spot_list)]
            if save_spots:
                with _held(spot_file_lock):
                    save_spots_to_fov_file(save_filename, spot_list, data_type, region_ids,
                                           raw_spot_list=raw_spot_list, overwrite=overwrite_spot, verbose=verbose)
    finally:
        for st in resident:
            if isinstance(st, L.DeviceStack):
                st.free()
    if return_spots:
        return spot_list, raw_spot_list
    return


# the keyword arguments of batch_process_image_to_spots and their defaults, in positional order (reference :60-87)
_BATCH_ARGS = ('dax_filename', 'sel_channels', 'save_filename', 'data_type', 'region_ids', 'ref_filename', 'load_file_lock',
               'warp_image', 'correction_args', 'save_image', 'empty_value', 'fov_savefile_lock', 'overwrite_image',
               'drift_args', 'save_drift', 'drift_filename', 'drift_file_lock', 'overwrite_drift', 'fit_spots', 'fit_in_mask',
               'fitting_args', 'save_spots', 'spot_file_lock', 'overwrite_spot', 'verbose', 'return_spots', 'fit_workers')
_BATCH_DEFAULTS = dict(load_file_lock=None, warp_image=True, correction_args={}, save_image=True, empty_value=0,
                       fov_savefile_lock=None, overwrite_image=False, drift_args={}, save_drift=True, drift_filename=None,
                       drift_file_lock=None, overwrite_drift=False, fit_spots=True, fit_in_mask=False, fitting_args={},
                       save_spots=True, spot_file_lock=None, overwrite_spot=False, verbose=False, return_spots=False,
                       fit_workers=None)


def _plan_key(kw):
    """Movies that can share one MoviePlan: same channels, same reference image and profile objects, same options."""
    def ident(v):
        if isinstance(v, dict):
            return tuple(sorted((str(k), ident(x)) for k, x in v.items()))
        if isinstance(v, (list, tuple)):
            return tuple(ident(x) for x in v)
        if isinstance(v, (str, int, float, bool, type(None))):
            return v
        if isinstance(v, np.ndarray) and v.size <= 64:
            return (v.shape, v.dtype.str, v.tobytes())
        return ('obj', id(v))   # profiles, reference image: shared by reference
    return (tuple(str(c) for c in kw['sel_channels']), ident(kw['ref_filename']), bool(kw['warp_image']),
            ident(kw['correction_args']), ident(kw['drift_args']), ident(kw['fitting_args']), bool(kw['verbose']),
            bool(kw['save_image']), bool(kw['fit_spots']), kw['data_type'])


def _pipelined_movies(tasks, chunk=8):
    """``batch_process_image_to_spots`` for many movies through ``ia3_process_movies`` (io_tools.load.MoviePlan): movie k+1
    is uploaded while the corrections, drift and warps of the movies before it run, and the channels of several movies
    are fitted by one group fitter.  ``tasks``: full keyword dicts.  Returns {task index: return value} for the movies it
    took; the rest — anything the pipelined entry does not cover, movies whose save file already holds some of their
    images — is left to the per-movie path.  File contents are those of the per-movie path."""
    from ..io_tools.load import MoviePlan
    done = {}
    groups = {}
    for i, kw in enumerate(tasks):
        try:
            if kw['fit_in_mask'] or kw['fit_workers'] not in (None, 1) or not kw['warp_image']:
                continue
            _check_batch_arguments(kw['dax_filename'], kw['save_filename'], kw['ref_filename'], kw['sel_channels'], kw['region_ids'])
            _require_type(kw['data_type'])
        except Exception:
            continue   # the per-movie call raises it in its own words
        groups.setdefault((_plan_key(kw), _dax_info(kw['dax_filename'])[:3]), []).append(i)
    for key, members in groups.items():
        kw0 = tasks[members[0]]
        channels = [str(c) for c in kw0['sel_channels']]
        # ---- what the save file holds (reference :121-166): only movies none of whose images can be reused -----------
        fresh, start, measure = [], {}, {}
        for i in members:
            kw = tasks[i]
            rids = [int(r) for r in kw['region_ids']]
            redo = kw['overwrite_image'] or kw['overwrite_drift']
            usable, drifts = False, []
            with _held(kw['fov_savefile_lock']), SaveFile(kw['save_filename'], kw['data_type']) as sf:
                for rid in rids:
                    slot = sf.row(rid)
                    drifts.append(sf['drifts'][slot, :])
                    if not redo and int(sf['flags'][slot]) - 1 == int(kw['warp_image']):
                        im = sf['ims'][slot]
                        usable = usable or bool(im.any() if kw['empty_value'] == 0 else (im != kw['empty_value']).any())
            if usable:
                continue
            fresh.append(i)
            start[i], measure[i] = _stored_drift(drifts, kw['overwrite_drift'])
        if not fresh:
            continue
        try:
            seed_th = {c: Channel_2_SeedTh[c] for c in channels} if kw0['fit_spots'] else None
            # correct_fov_image(..., **correction_args, **drift_args) in the per-movie call (reference :178-186)
            plan = MoviePlan(channels, ref_image=kw0['ref_filename'], calculate_drift=any(measure.values()),
                             warp_image=True, verbose=kw0['verbose'], seed_th=seed_th, fitting_args=kw0['fitting_args'],
                             fit_spots=kw0['fit_spots'], frames=_dax_info(kw0['dax_filename'])[0],
                             **kw0['correction_args'], **kw0['drift_args'])
        except (NotImplementedError, KeyError, TypeError):
            continue
        for c0 in range(0, len(fresh), chunk):
            part = fresh[c0:c0 + chunk]
            res = plan.run([tasks[i]['dax_filename'] for i in part], drifts_in=[start[i] for i in part],
                           measure_drift=[measure[i] for i in part], want_images=kw0['save_image'])
            for i, r in zip(part, res):
                kw = tasks[i]
                rids = [int(x) for x in kw['region_ids']]
                drift = r['drift'] if measure[i] else np.array(start[i], dtype=np.float32)
                if kw['save_image']:
                    with _held(kw['fov_savefile_lock']):
                        save_image_to_fov_file(kw['save_filename'], r['images'], kw['data_type'], rids, True, drift,
                                               r['drift_flag'], kw['overwrite_image'], kw['verbose'])
                spot_list, raw_spot_list = np.array([]), []
                if kw['fit_spots']:
                    # fit_fov_image returns np.array([]) for an image without seeds (spot_tools/fitting.py:206-207)
                    raw_spot_list = [t if ns > 0 else np.array([]) for t, ns in zip(r['tables'], r['n_seeds'])]
                    spot_list = [raw.copy() for raw in raw_spot_list]
                    if kw['save_spots']:
                        with _held(kw['spot_file_lock']):
                            save_spots_to_fov_file(kw['save_filename'], spot_list, kw['data_type'], rids,
                                                   raw_spot_list=raw_spot_list, overwrite=kw['overwrite_spot'],
                                                   verbose=kw['verbose'])
                done[i] = (spot_list, raw_spot_list) if kw['return_spots'] else None
    return done


def batch_process_images_to_spots(args_list, num_threads=4, shared_kwargs=None, pipeline=None):
    """The fan-out of ``Field_of_View._process_image_to_spots`` (classes/field_of_view.py:1015-1142) for one GPU.

    The reference starts ``mp.Pool(num_threads).starmap(batch_process_image_to_spots, args)`` with manager locks for
    the save file; every task pickles its arguments (reference image and profiles included).  Here the movies go through
    ONE pipelined library call (``ia3_process_movies``, see ``_pipelined_movies``): upload of the next movie, corrections /
    drift / warps of the current ones and cross-movie group fits overlap on library threads, profiles and the reference
    bead image are uploaded once.  Movies that entry does not cover (options outside it, save files that already hold
    some of their images) — or all of them with ``pipeline=False`` — run as threads of this process, each with its own HIP
    streams, a plain ``threading.RLock`` serialising the save file.  ``args_list``: one dict of
    ``batch_process_image_to_spots`` keyword arguments per movie (or a tuple of its positional arguments);
    ``shared_kwargs`` are added to each.  ``pipeline``: None = use it when there is more than one movie.  Returns the
    per-movie return values, in order.  Across GPUs: one such process per device over ``parallel.shard_fovs``."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    shared_kwargs = dict(shared_kwargs or {})
    file_lock = threading.RLock()   # images and spot tables live in the same file: one lock for both
    L.lib()   # load the library before the threads start

    def run(task):
        kw = dict(shared_kwargs)
        pos = ()
        if isinstance(task, dict):
            kw.update(task)
        else:
            pos = tuple(task)
        kw.setdefault('fov_savefile_lock', file_lock)
        kw.setdefault('spot_file_lock', file_lock)
        for name in ('fitting_args', 'correction_args', 'drift_args'):   # written to inside the call: private copies
            if name in kw:
                kw[name] = dict(kw[name])
        return batch_process_image_to_spots(*pos, **kw)

    results = {}
    use_pipeline = (len(args_list) > 1 and num_threads > 1) if pipeline is None else bool(pipeline)
    if use_pipeline:
        full = []
        for task in args_list:
            kw = dict(_BATCH_DEFAULTS)
            kw.update(shared_kwargs)
            if isinstance(task, dict):
                kw.update(task)
            else:
                kw.update(dict(zip(_BATCH_ARGS, task)))
            if kw.get('fov_savefile_lock') is None:
                kw['fov_savefile_lock'] = file_lock
            if kw.get('spot_file_lock') is None:
                kw['spot_file_lock'] = file_lock
            for name in ('fitting_args', 'correction_args', 'drift_args'):
                kw[name] = dict(kw[name])
            full.append(kw if all(k in kw for k in _BATCH_ARGS[:6]) else None)
        if all(kw is not None for kw in full):
            results = _pipelined_movies(full)
    rest = [i for i in range(len(args_list)) if i not in results]
    if num_threads <= 1 or len(rest) <= 1:
        for i in rest:
            results[i] = run(args_list[i])
    else:
        with ThreadPoolExecutor(max_workers=int(num_threads)) as pool:
            for i, r in zip(rest, pool.map(run, [args_list[i] for i in rest])):
                results[i] = r
    return [results[i] for i in range(len(args_list))]
