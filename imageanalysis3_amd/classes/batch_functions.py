"""classes/batch_functions.py twin — the per-image production entry and its save-file helpers.

``batch_process_image_to_spots`` (reference :60-303) is what ``Field_of_View._process_image_to_spots`` hands to its
process pool for every (round folder, FOV) movie: look up what the FOV save file already holds, run
``correct_fov_image`` for the channels still missing, store images / drift / warp flags, fit every channel with
``fit_fov_image`` and store ``spots`` / ``raw_spots``.  Here the corrected channels stay resident on the GPU between
the two halves (``correct_fov_image(..., return_device=True)`` -> ``fit_fov_image``); the host copy is only made for
the save file.  File layout, update rules and return values follow the reference line by line; HDF5 access goes
through ``io_tools.h5lite`` (libhdf5 via ctypes — this image has no h5py for the system interpreter).

``create_fov_save_file`` restates the data-type group that ``Field_of_View._save_to_file`` creates
(classes/field_of_view.py:1314-1398), so that a save file can be made without the (out-of-scope) Field_of_View class.
"""
import os
import pickle
import time
import numpy as np
from scipy import ndimage

from . import _allowed_kwds, _max_num_seeds
from .. import _image_dtype
from ..io_tools import h5lite as h5py
from ..io_tools.load import correct_fov_image
from ..spot_tools.fitting import fit_fov_image, get_centers  # noqa: F401  (re-exported as in the reference)
from .. import _lib as L

# reference :10-17
Channel_2_SeedTh = {
    '750': 600,
    '647': 600,
    '561': 600,
    '748': 1000,
    '637': 1000,
    '545': 1000,
}


def create_fov_save_file(filename, data_type, ids, channels, single_im_size, max_num_seeds=None, overwrite=False):
    """The ``data_type`` group of a FOV save file as classes/field_of_view.py:1314-1398 lays it out:
    ``ids (n,) i4``, ``channels (n,) S3``, ``ims (n,Z,X,Y) u2`` chunked per image, ``spots`` / ``raw_spots``
    ``(n, L, 11) f4`` growable along axis 1 (L = max_num_seeds, default classes._max_num_seeds), ``drifts (n,3) f4``,
    ``flags (n,) u1`` (0 empty, 1 unwarped image, 2 warped image).  Existing members are kept."""
    if data_type not in _allowed_kwds:
        raise ValueError(f"Wrong input data_type:{data_type}, should be among {_allowed_kwds}.")
    if len(ids) != len(channels):
        raise ValueError("ids and channels should have the same length")
    _n = len(ids)
    _im_shape = (int(_n),) + tuple(int(_s) for _s in single_im_size)
    _chunk_shape = (1,) + tuple(int(_s) for _s in single_im_size)
    _spot_save_len = int(_max_num_seeds if max_num_seeds is None else max_num_seeds)
    with h5py.File(filename, "w" if overwrite else "a", libver='latest') as _f:
        _grp = _f.require_group(data_type)
        if 'ids' not in _grp:
            _grp.create_dataset('ids', (_n,), dtype='i', data=np.array(ids, dtype=np.int32))
        if 'channels' not in _grp:
            _grp.create_dataset('channels', (_n,), dtype='S3', data=[str(_ch).encode('utf8') for _ch in channels])
        if 'ims' not in _grp:
            _grp.create_dataset('ims', _im_shape, dtype='u2', chunks=_chunk_shape)
        for _name in ('spots', 'raw_spots'):
            if _name not in _grp:
                _grp.create_dataset(_name, (_n, _spot_save_len, 11), dtype='f', maxshape=(_n, None, 11), chunks=True)
        if 'drifts' not in _grp:
            _grp.create_dataset('drifts', (_n, 3), dtype='f')
        if 'flags' not in _grp:
            _grp.create_dataset('flags', (_n,), dtype='u1')
    return filename


def save_image_to_fov_file(filename, ims, data_type, region_ids,
                           warp_image=False, drift=None, drift_flag=None,
                           overwrite=False, verbose=True):
    """reference :305-368 — write images (and their drift) into the slots of ``region_ids`` that are still empty
    (flag 0) or when ``overwrite``; flag 1 = stored unwarped, 2 = stored warped.  Returns whether anything was written."""
    if not os.path.isfile(filename):
        raise IOError(f"save file: {filename} doesn't exist!")
    if data_type not in _allowed_kwds:
        raise ValueError(f"Wrong input data_type:{data_type}, should be among {_allowed_kwds}.")
    if len(ims) != len(region_ids):
        raise ValueError(f"Wrong input region_ids:{region_ids}, should of same length as ims, len={len(ims)}.")
    if drift is not None:
        if len(np.shape(drift)) == 1:
            _all_drifts = [drift for _im in ims]
        elif len(drift) == len(ims):
            _all_drifts = drift
        else:
            raise IndexError(f"Length of drift should match ims")
    if verbose:
        print(f"- writting {data_type} info to file:{filename}")
        _save_start = time.time()
    _updated_ims = []
    _updated_drifts = []
    _saving_flag = False
    with h5py.File(filename, "a", libver='latest') as _f:
        _grp = _f.require_group(data_type)
        for _i, (_id, _im) in enumerate(zip(region_ids, ims)):
            _index = list(_grp['ids'][:]).index(_id)
            _flag = _grp['flags'][_index]
            if _flag == 0 or overwrite:
                _saving_flag = True
                _grp['ims'][_index] = _im
                if not warp_image:
                    _grp['flags'][_index] = 1
                else:
                    _grp['flags'][_index] = 2
                _updated_ims.append(_id)
                if drift is not None:
                    _grp['drifts'][_index, :] = _all_drifts[_i]
                    _updated_drifts.append(_id)
    if verbose:
        if _saving_flag:
            print(f"-- updated ims for id:{_updated_ims}, drifts for id:{_updated_drifts} in {time.time()-_save_start:.3f}s")
        else:
            print(f"-- images and drifts already exist, skip.")
    return _saving_flag


def load_image_from_fov_file(filename, data_type, region_ids,
                             image_dtype=_image_dtype, load_drift=False, verbose=True):
    """reference :371-419 — images and warp flags (and drifts) of ``region_ids``, in the order given."""
    if not os.path.isfile(filename):
        raise IOError(f"load file: {filename} doesn't exist!")
    if data_type not in _allowed_kwds:
        raise ValueError(f"Wrong input data_type:{data_type}, should be among {_allowed_kwds}.")
    if isinstance(region_ids, (int, np.integer)):
        _region_ids = [int(region_ids)]
    elif isinstance(region_ids, list) or isinstance(region_ids, np.ndarray):
        _region_ids = [int(_id) for _id in region_ids]
    else:
        raise TypeError(f"Wrong input type for region_ids:{region_ids}")
    if verbose:
        print(f"- loading {data_type} info from file:{os.path.basename(filename)}", end=' ')
        _load_start = time.time()
    _ims = []
    _flags = []
    if load_drift:
        _drifts = []
    with h5py.File(filename, "a", libver='latest') as _f:
        _grp = _f[data_type]
        for _i, _id in enumerate(_region_ids):
            _index = list(_grp['ids'][:]).index(_id)
            _ims.append(_grp['ims'][_index])
            _flags.append(_grp['flags'][_index])
            if load_drift:
                _drifts.append(_grp['drifts'][_index, :])
    if verbose:
        print(f"in {time.time()-_load_start:.3f}s.")
    if load_drift:
        return _ims, _flags, _drifts
    else:
        return _ims, _flags


def save_spots_to_fov_file(filename, spot_list, data_type, region_ids,
                           raw_spot_list=None,
                           overwrite=False, verbose=True):
    """reference :422-493 — write fitted spots (and the un-translated ``raw_spots``) into the rows of ``region_ids``
    whose stored table is still all zero (or when ``overwrite``), growing the tables along axis 1 when a list is
    longer than what is stored."""
    if not os.path.isfile(filename):
        raise IOError(f"save file: {filename} doesn't exist!")
    if data_type not in _allowed_kwds:
        raise ValueError(f"Wrong input data_type:{data_type}, should be among {_allowed_kwds}.")
    if len(spot_list) != len(region_ids):
        raise ValueError(f"Wrong input region_ids:{region_ids}, should of same length as spots, len={len(spot_list)}.")
    if raw_spot_list is not None and len(raw_spot_list) != len(spot_list):
        raise IndexError(f"length of input spot_list and raw_spot list should match, {len(spot_list)}, {len(raw_spot_list)}")
    if verbose:
        print(f"- writting {data_type} spots into file:{filename}")
        _save_start = time.time()
    _updated_spots = []
    with h5py.File(filename, "a", libver='latest') as _f:
        _grp = _f[data_type]
        for _i, (_id, _spots) in enumerate(zip(region_ids, spot_list)):
            _saved_shape = _grp['spots'].shape
            _max_shape = _grp['spots'].maxshape
            # not large enough even with maxshape: recreate the tables
            if _saved_shape[1] < len(_spots) and _max_shape[1] is not None and _max_shape[1] < len(_spots):
                if verbose:
                    print(f"-- recreate {data_type}_spots and {data_type}_raw_spots from {_saved_shape[1]} to {len(_spots)}.")
                _existing_spots = _grp['spots'][:]
                _existing_raw_spots = _grp['raw_spots'][:]
                del(_grp['spots'])
                del(_grp['raw_spots'])
                _grp.create_dataset('spots',
                                    (_saved_shape[0], len(_spots), _saved_shape[2]),
                                    dtype='f', maxshape=(_saved_shape[0], None, _saved_shape[2]), chunks=True)
                _grp['spots'][:, :_saved_shape[1], :] = _existing_spots
                _grp.create_dataset('raw_spots',
                                    (_saved_shape[0], len(_spots), _saved_shape[2]),
                                    dtype='f', maxshape=(_saved_shape[0], None, _saved_shape[2]), chunks=True)
                _grp['raw_spots'][:, :_saved_shape[1], :] = _existing_raw_spots
            # maxshape allows it: resize
            elif _saved_shape[1] < len(_spots):
                if verbose:
                    print(f"-- resize {data_type}_spots and {data_type}_raw_spots from {_saved_shape[1]} to {len(_spots)}.")
                _grp['spots'].resize(len(_spots), 1)
                _grp['raw_spots'].resize(len(_spots), 1)

            _index = list(_grp['ids'][:]).index(_id)
            if np.sum(_grp['spots'][_index]) == 0 or overwrite:
                _grp['spots'][_index, :len(_spots), :] = _spots
                _updated_spots.append(_id)
            if 'raw_spots' in _grp.keys():
                if np.sum(_grp['raw_spots'][_index]) == 0 or overwrite:
                    _grp['raw_spots'][_index, :len(raw_spot_list[_i]), :] = raw_spot_list[_i]
    if verbose:
        print(f"-- updated spots for id:{_updated_spots} in {time.time()-_save_start:.3f}s")
    return True


def _drift_key(image_filename):
    return os.path.join(os.path.basename(os.path.dirname(image_filename)), os.path.basename(image_filename))


def save_drift_to_file(drift_filename, image_filename, drift, overwrite=False, verbose=True):
    """reference :496-519 — pickled dict ``{'<folder>/<movie>.dax': drift}``."""
    if os.path.isfile(drift_filename):
        drift_dict = pickle.load(open(drift_filename, 'rb'))
    else:
        drift_dict = {}
    _update = False
    _key = _drift_key(image_filename)
    if _key not in drift_dict or overwrite:
        drift_dict[_key] = drift
        _update = True
    if _update:
        if verbose:
            print(f"-- update drift of {_key} into file:{drift_filename}")
        pickle.dump(drift_dict, open(drift_filename, 'wb'))
    else:
        if verbose:
            print(f"-- no updates in drift, skip.")
    return True


def create_drift_file(drift_filename, ref_filename,
                      n_dim=3,
                      overwrite=False, verbose=True):
    """reference :523-556 — start the drift dict with a zero drift for the reference movie."""
    if os.path.isfile(drift_filename) and not overwrite:
        drift_dict = pickle.load(open(drift_filename, 'rb'))
    else:
        drift_dict = {}
    _ref_key = _drift_key(ref_filename)
    if _ref_key not in drift_dict:
        drift_dict[_ref_key] = np.zeros(n_dim)
        _update = True
    else:
        _update = False
    if _update:
        if not os.path.isdir(os.path.dirname(drift_filename)):
            if verbose:
                print(f"--- creating folder:{os.path.dirname(drift_filename)}")
            os.makedirs(os.path.dirname(drift_filename))
        if verbose:
            print(f"-- create drift file:{drift_filename} with reference:{_ref_key}")
        pickle.dump(drift_dict, open(drift_filename, 'wb'))
    else:
        if verbose:
            print(f"-- no updates in drift file:{drift_filename}, skip.")
    return True


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
                                 return_spots=False):
    """reference :60-303 — one movie: corrected images + drift into the FOV save file, then spots of every selected
    channel.  Returns None like the reference (``return_spots=True``, an extension, returns ``(spots, raw_spots)``)."""
    ## check inputs (:92-118)
    if not os.path.isfile(dax_filename):
        raise IOError(f"Dax file: {dax_filename} is not a file, exit!")
    if not isinstance(dax_filename, str) or dax_filename[-4:] != '.dax':
        raise IOError(f"Dax file: {dax_filename} has wrong data type, exit!")
    sel_channels = [str(ch) for ch in sel_channels]
    if verbose:
        print(f"+ batch process image: {dax_filename} for channels:{sel_channels}")
    if not os.path.isfile(save_filename):
        raise IOError(f"HDF5 file: {save_filename} is not a file, exit!")
    if not isinstance(save_filename, str) or save_filename[-5:] != '.hdf5':
        raise IOError(f"HDF5 file: {save_filename} has wrong data type, exit!")
    if isinstance(ref_filename, str):
        if not os.path.isfile(ref_filename):
            raise IOError(f"Dax file: {ref_filename} is not a file, exit!")
        elif ref_filename[-4:] != '.dax':
            raise IOError(f"Dax file: {ref_filename} has wrong data type, exit!")
    elif isinstance(ref_filename, np.ndarray):
        pass
    else:
        raise TypeError(f"ref_filename should be np.ndarray or string of path, but {type(ref_filename)} is given")
    if len(region_ids) != len(sel_channels):
        raise ValueError(f"Wrong input region_ids:{region_ids}, should of same length as sel_channels:{sel_channels}.")
    region_ids = [int(_id) for _id in region_ids]

    ## what does the save file hold already (:121-166)
    if fov_savefile_lock is not None:
        fov_savefile_lock.acquire()
    _ims, _warp_flags, _drifts = load_image_from_fov_file(save_filename,
                                                          data_type, region_ids,
                                                          load_drift=True,
                                                          verbose=verbose)
    if fov_savefile_lock is not None:
        fov_savefile_lock.release()
    _process_flags = []
    _process_sel_channels = []
    _carryover_ims = []
    _carryover_sel_channels = []
    for _im, _flg, _drift, _rid, _ch in zip(_ims, _warp_flags, _drifts, region_ids, sel_channels):
        if overwrite_image or overwrite_drift:
            _process_flags.append(True)
            _process_sel_channels.append(_ch)
        else:
            if (_im != empty_value).any() and _flg - 1 == int(warp_image):
                _process_flags.append(False)
                _carryover_ims.append(_im.copy())
                _carryover_sel_channels.append(_ch)
            else:
                _process_flags.append(True)
                _process_sel_channels.append(_ch)
    del(_ims)
    _process_drift = list(set([tuple(_dft) for _dft in _drifts]))
    if len(_process_drift) == 1 and np.array(_process_drift[0]).any() and not overwrite_drift:
        _process_drift = np.array(_process_drift[0])     # one unique non-zero drift stored: use it
        _corr_drift = False
    else:
        _process_drift = np.zeros(len(_process_drift[0]))
        _corr_drift = True

    ## correct the images still missing (:169-206); they stay on the device for the fit
    _resident = []   # DeviceStacks to release at the end
    try:
        if np.sum(_process_flags) > 0:
            if verbose:
                print(f"-- {_process_sel_channels} images are required to process, {_carryover_sel_channels} images are loaded from save file: {save_filename}")
            if warp_image:
                _processed_ims, _drift, _drift_flag = correct_fov_image(
                    dax_filename,
                    _process_sel_channels,
                    load_file_lock=load_file_lock,
                    calculate_drift=_corr_drift,
                    drift=_process_drift,
                    ref_filename=ref_filename,
                    warp_image=warp_image,
                    return_drift=True, verbose=verbose, return_device=True,
                    **correction_args, **drift_args)
            else:
                _processed_ims, _processed_warp_funcs, _drift, _drift_flag = correct_fov_image(
                    dax_filename,
                    _process_sel_channels,
                    load_file_lock=load_file_lock,
                    calculate_drift=_corr_drift,
                    drift=_process_drift,
                    ref_filename=ref_filename,
                    warp_image=warp_image,
                    return_drift=True, verbose=verbose, return_device=True,
                    **correction_args, **drift_args)
            _resident = list(_processed_ims)
        else:
            _processed_ims = []
            if not warp_image:
                _processed_warp_funcs = []
            _drift = np.array(_process_drift)
            _drift_flag = 0

        ## merge processed and carried-over images (:209-229)
        _processed_ims = list(_processed_ims)
        _sel_ims = []       # what fit_fov_image gets: resident stack or host array
        for _ch, _flg in zip(sel_channels, _process_flags):
            if not _flg:
                _sel_ims.append(_carryover_ims.pop(0))
            else:
                _sel_ims.append(_processed_ims.pop(0))
        if not warp_image:
            _warp_funcs = []
            for _ch, _flg in zip(sel_channels, _process_flags):
                if not _flg:
                    from ..correction_tools.chromatic import generate_chromatic_function
                    _warp_funcs.append(
                        generate_chromatic_function(correction_args['chromatic_profile'][str(_ch)], _drift)
                    )
                else:
                    _warp_funcs.append(
                        _processed_warp_funcs.pop(0)
                    )

        ## save images + drift (:232-245)
        if save_image:
            _host_ims = [_im.download() if isinstance(_im, L.DeviceStack) else _im for _im in _sel_ims]
            if fov_savefile_lock is not None:
                fov_savefile_lock.acquire()
            _save_img_success = save_image_to_fov_file(
                save_filename, _host_ims, data_type, region_ids,
                warp_image, _drift, _drift_flag,
                overwrite_image, verbose)
            if fov_savefile_lock is not None:
                fov_savefile_lock.release()
            del(_host_ims)

        ## fit (:248-300)
        _raw_spot_list = []
        if fit_spots:
            if fit_in_mask:
                if 'seed_mask' not in fitting_args or fitting_args['seed_mask'] is None:
                    raise KeyError(f"seed_mask should be given if fit_in_mask specified")
                if warp_image:
                    _shifted_mask = fitting_args['seed_mask']
                else:
                    if verbose:
                        print(f"-- start traslating seed_mask by drift: {_drift}", end=' ')
                        _translate_start = time.time()
                    _shifted_mask = ndimage.shift(fitting_args['seed_mask'],
                                                  -_drift,
                                                  mode='constant',
                                                  cval=0)
                fitting_args['seed_mask'] = _shifted_mask
                if verbose:
                    print(f"-- in {time.time()-_translate_start:.2f}s.")
                    _translate_start = time.time()
            _spot_list = []
            for _ich, (_im, _ch) in enumerate(zip(_sel_ims, sel_channels)):
                fitting_args['th_seed'] = Channel_2_SeedTh[str(_ch)]
                _raw_spots = fit_fov_image(
                    _im, _ch, verbose=verbose,
                    **fitting_args,
                )
                if not warp_image:
                    _func = _warp_funcs[_ich]
                    _spots = _func(_raw_spots)
                else:
                    _spots = _raw_spots.copy()
                _spot_list.append(_spots)
                _raw_spot_list.append(_raw_spots)
            if save_spots:
                if spot_file_lock is not None:
                    spot_file_lock.acquire()
                _save_spt_success = save_spots_to_fov_file(
                    save_filename, _spot_list, data_type, region_ids,
                    raw_spot_list=_raw_spot_list,
                    overwrite=overwrite_spot, verbose=verbose)
                if spot_file_lock is not None:
                    spot_file_lock.release()
        else:
            _spot_list = np.array([])
    finally:
        for _st in _resident:
            if isinstance(_st, L.DeviceStack):
                _st.free()
    if return_spots:
        return _spot_list, _raw_spot_list
    return


def batch_process_images_to_spots(args_list, num_threads=4, shared_kwargs=None):
    """The fan-out of ``Field_of_View._process_image_to_spots`` (classes/field_of_view.py:1015-1142) for one GPU.

    The reference starts ``mp.Pool(num_threads).starmap(batch_process_image_to_spots, args)`` with manager locks for
    the save file; every task pickles its arguments (reference image and profiles included).  Here the tasks are
    threads of the one process that owns the GPU: libia3 gives each thread its own HIP streams, so the corrections,
    warps and fits of different movies overlap on the device, profiles / the reference bead image are shared by
    reference (hand them over as ``DeviceBuffer`` / ndarray once), and plain ``threading.Lock`` objects serialise the
    save file.  ``args_list``: one dict of ``batch_process_image_to_spots`` keyword arguments per movie (or a tuple of
    its positional arguments); ``shared_kwargs`` are added to each.  Returns the per-movie return values, in order.
    Across GPUs: one such process per device over ``parallel.shard_fovs``."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    shared_kwargs = dict(shared_kwargs or {})
    _file_lock = threading.RLock()   # images and spot tables live in the same file: one lock for both
    _spot_lock = _file_lock
    L.lib()   # load the library before the threads start

    def _run(_args):
        if isinstance(_args, dict):
            _kw = dict(shared_kwargs)
            _kw.update(_args)
            _pos = ()
        else:
            _pos, _kw = tuple(_args), dict(shared_kwargs)
        _kw.setdefault('fov_savefile_lock', _file_lock)
        _kw.setdefault('spot_file_lock', _spot_lock)
        # the argument dicts are written to inside the call (th_seed, seed_mask): give every task its own copy
        for _k in ('fitting_args', 'correction_args', 'drift_args'):
            if _k in _kw:
                _kw[_k] = dict(_kw[_k])
        return batch_process_image_to_spots(*_pos, **_kw)

    if num_threads <= 1 or len(args_list) <= 1:
        return [_run(_a) for _a in args_list]
    with ThreadPoolExecutor(max_workers=int(num_threads)) as _pool:
        return list(_pool.map(_run, args_list))
