"""reference: io_tools/load.py — only ``find_image_background`` (:642-687) so far.

Host-side NumPy (a few thousand voxels per call).  SURVEY.md §8(f4) lists the batched
histogram-mode kernel for ``normalize_local`` as a "next" row; until it exists this helper is
the reference's own arithmetic on the host, outside the measured hot path.
"""
import ctypes as C
import numpy as np
from .. import _image_dtype
from .. import _lib as L


def illumination_correction(im, profile, output_dtype=np.uint16):
    """io_tools/load.py:373-384 — ``(im.astype(np.float32) / profile[np.newaxis,:]).astype(output_dtype)``
    on the device.  ``im`` (Z,X,Y) uint16, ``profile`` (X,Y) float32 or float64."""
    a = np.ascontiguousarray(im)
    if a.dtype != np.uint16 or np.dtype(output_dtype) != np.uint16:
        raise TypeError("illumination_correction takes and returns uint16 stacks")
    p = np.ascontiguousarray(profile)
    if p.dtype not in (np.float32, np.float64):
        p = p.astype(np.float64)
    if p.shape != a.shape[1:]:
        raise IndexError(f"illumination profile shape {p.shape} should be {a.shape[1:]}")
    out = np.empty_like(a)
    L.check(L.lib().ia3_illumination_correct(L.ptr(a), a.shape[0], a.shape[1], a.shape[2], L.ptr(p),
                                             1 if p.dtype == np.float32 else 2, L.ptr(out)))
    return out


def bleedthrough_correction(ims, bleed_profile, output_dtype=np.uint16):
    """io_tools/load.py:348-370 — ``new_i = sum_j ims[j] * bleed_profile[i, j]`` clipped to uint16, on the
    device.  ``ims``: list of C (Z,X,Y) uint16 stacks (the corr_channels), ``bleed_profile`` (C,C,X,Y)."""
    ims = [np.ascontiguousarray(_im) for _im in ims]
    n_ch = len(ims)
    if any(_im.dtype != np.uint16 for _im in ims) or np.dtype(output_dtype) != np.uint16:
        raise TypeError("bleedthrough_correction takes and returns uint16 stacks")
    p = np.ascontiguousarray(bleed_profile)
    if p.dtype not in (np.float32, np.float64):
        p = p.astype(np.float64)
    if p.shape != (n_ch, n_ch) + ims[0].shape[1:]:
        raise IndexError(f"bleed_profile shape {p.shape} should be {(n_ch, n_ch) + ims[0].shape[1:]}")
    outs = [np.empty_like(_im) for _im in ims]
    arr_in = (C.c_void_p * n_ch)(*[_im.ctypes.data for _im in ims])
    arr_out = (C.c_void_p * n_ch)(*[_o.ctypes.data for _o in outs])
    Z, X, Y = ims[0].shape
    L.check(L.lib().ia3_bleedthrough_correct(arr_in, n_ch, Z, X, Y, L.ptr(p), 1 if p.dtype == np.float32 else 2,
                                             arr_out))
    return outs


def _background_edges(dtype, bin_size):
    """The reference's bin edges (io_tools/load.py:655-658) as the float64 vector the device entry takes."""
    info = np.iinfo(dtype)
    return np.ascontiguousarray(np.arange(info.min, info.max, bin_size), dtype=np.float64)


def _stack3(im):
    if not isinstance(im, np.ndarray):
        raise TypeError("im should be a numpy.ndarray")
    a = im if im.ndim == 3 else im.reshape((1,) * (3 - im.ndim) + im.shape) if im.ndim < 3 else im.reshape(-1, *im.shape[-2:])
    return L.as_stack_array(a)


def load_correction_profile(corr_type, corr_channels=None,
                            correction_folder=None, all_channels=None,
                            ref_channel='647', im_size=None, verbose=False):
    """io_tools/load.py:553-640 — read a correction profile from the correction folder:
    ``bleedthrough_correction_<chs high→low>_<X>_<Y>.npy`` -> (C,C,X,Y) array;
    ``chromatic_correction_<ch>_<ref>_<Z>_<X>_<Y>.npy`` -> dict of dense fields (None for the reference channel);
    ``chromatic_correction_<ch>_<ref>_<Z>_<X>_<Y>_const.pkl`` -> dict of polynomial constants;
    ``illumination_correction_<ch>_<X>x<Y>.npy`` -> dict of (X,Y) maps."""
    import os
    import pickle
    from .. import _corr_channels, _correction_folder, _allowed_colors, _image_size
    corr_channels = _corr_channels if corr_channels is None else corr_channels
    correction_folder = _correction_folder if correction_folder is None else correction_folder
    all_channels = _allowed_colors if all_channels is None else all_channels
    im_size = _image_size if im_size is None else im_size
    _allowed_types = ['chromatic', 'illumination', 'bleedthrough', 'chromatic_constants']
    _type = str(corr_type).lower()
    if _type not in _allowed_types:
        raise ValueError(f"Wrong input corr_type, should be one of {_allowed_types}")
    _all_channels = [str(_ch) for _ch in all_channels]
    _corr_channels_ = [str(_ch) for _ch in corr_channels]
    for _channel in _corr_channels_:
        if _channel not in _all_channels:
            raise ValueError(f"Wrong input channel:{_channel}, should be one of {_all_channels}")
    _ref_channel = str(ref_channel).lower()
    if _ref_channel not in _all_channels:
        raise ValueError(f"Wrong input ref_channel:{_ref_channel}, should be one of {_all_channels}")
    if verbose:
        print(f"-- loading {_type} correction profile from file", end=':')
    if _type == 'bleedthrough':
        _basename = _type + '_correction' \
            + '_' + '_'.join(sorted(_corr_channels_, key=lambda v: -int(v))) \
            + '_' + str(im_size[-2]) + '_' + str(im_size[-1]) + '.npy'
        if verbose:
            print(_basename)
        _pf = np.load(os.path.join(correction_folder, _basename), allow_pickle=True)
        _pf = _pf.reshape(len(_corr_channels_), len(_corr_channels_), im_size[-2], im_size[-1])
    elif _type == 'chromatic' or _type == 'chromatic_constants':
        if verbose:
            print('')
        _pf = {}
        for _channel in _corr_channels_:
            if _channel != _ref_channel:
                _basename = 'chromatic_correction' + '_' + str(_channel) + '_' + str(_ref_channel)
                for _d in im_size:
                    _basename += f'_{int(_d)}'
                _basename += '.npy' if _type == 'chromatic' else '_const.pkl'
                if verbose:
                    print('\t', _channel, _basename)
                if _type == 'chromatic':
                    _pf[_channel] = np.load(os.path.join(correction_folder, _basename), allow_pickle=True)
                else:
                    _pf[_channel] = pickle.load(open(os.path.join(correction_folder, _basename), 'rb'))
            else:
                if verbose:
                    print('\t', _channel, None)
                _pf[_channel] = None
    elif _type == 'illumination':
        if verbose:
            print('')
        _pf = {}
        for _channel in _corr_channels_:
            _basename = _type + '_correction' + '_' + str(_channel) \
                + '_' + str(im_size[-2]) + 'x' + str(im_size[-1]) + '.npy'
            if verbose:
                print('\t', _channel, _basename)
            _pf[_channel] = np.load(os.path.join(correction_folder, _basename), allow_pickle=True)
    return _pf


def find_image_background(im, dtype=_image_dtype, bin_size=10, make_plot=False, max_iter=10):
    """Histogram-peak background level (io_tools/load.py:642-687), computed by ``ia3_find_background(_dev)``
    (background.hip).  ``im``: uint16/float32 ndarray or a resident ``DeviceStack``.  ``make_plot`` is accepted
    for signature compatibility and ignored (plotting is out of scope)."""
    if dtype is None:
        dtype = im.dtype
    edges = _background_edges(dtype, bin_size)
    out = C.c_double(0.0)
    if isinstance(im, L.DeviceStack):
        L.check(L.lib().ia3_find_background_dev(im._h, L.dptr(edges), len(edges), int(max_iter), C.byref(out)))
    else:
        a = _stack3(im)
        L.check(L.lib().ia3_find_background(L.ptr(a), L.dtype_code(a), a.shape[0], a.shape[1], a.shape[2],
                                            L.dptr(edges), len(edges), int(max_iter), C.byref(out)))
    return np.float64(out.value)


def find_local_backgrounds(stack, centers_zxy, crop_size, dtype=_image_dtype, bin_size=10, make_plot=False,
                           max_iter=10):
    """``[find_image_background(im[generate_neighboring_crop(c, crop_size).to_slices()], ...) for c in centers]``
    (spot_tools/fitting.py:249-256) in one launch on a resident stack."""
    if dtype is None:
        dtype = stack.dtype
    edges = _background_edges(dtype, bin_size)
    c = np.ascontiguousarray(centers_zxy, dtype=np.float32).reshape(-1, 3)
    out = np.empty(len(c), dtype=np.float64)
    L.check(L.lib().ia3_local_background_dev(stack._h, L.ptr(c), len(c), int(crop_size), L.dptr(edges), len(edges),
                                             int(max_iter), L.dptr(out)))
    return out


# ---------------------------------------------------------------------------------------------------------------
# correct_fov_image (io_tools/load.py:166-522): the whole per-image chain on stacks that stay in HBM
# ---------------------------------------------------------------------------------------------------------------

def _dax_info(dax_filename):
    """(frames, X, Y, big_endian) from the .inf next to a .dax movie (visual_tools.py:990-1030)."""
    import os
    import re
    inf = os.path.splitext(dax_filename)[0] + ".inf"
    h = w = n = None
    big = False
    with open(inf, "r") as f:
        for line in f:
            m = re.match(r'frame dimensions = ([\d]+) x ([\d]+)', line)
            if m:
                h, w = int(m.group(1)), int(m.group(2))
            m = re.match(r'number of frames = ([\d]+)', line)
            if m:
                n = int(m.group(1))
            m = re.search(r' (big|little) endian', line)
            if m:
                big = m.group(1) == "big"
    if not h:
        h = w = 256
    if n is None:
        n = os.path.getsize(dax_filename) // (2 * h * w)
    return n, w, h, big


def read_dax(dax_filename):
    """The raw movie of a .dax/.inf pair as a (frames, width, height) uint16 array — what the reference's
    ``DaxReader(dax_filename).loadAll()`` returns (visual_tools.py:974-1083; host file I/O only)."""
    n, w, h, big = _dax_info(dax_filename)
    data = np.fromfile(dax_filename, dtype='>u2' if big else '<u2', count=n * h * w)
    return np.ascontiguousarray(data.reshape(n, w, h).astype(np.uint16, copy=False))


def load_dax_resident(dax_filename):
    """The same movie as a resident ``DeviceStack`` without a host copy: the file is streamed through pinned staging
    buffers (``ia3_stack_load_file``), file read and PCIe transfer overlapped."""
    n, w, h, big = _dax_info(dax_filename)
    return L.DeviceStack.from_file(dax_filename, n, w, h, big_endian=big)


def get_num_frame(dax_filename, frame_per_color=None, buffer_frame=10, empty_frame=0, verbose=False):
    """io_tools/load.py:17-45 — ([frames, dx, dy], number of colours) from the .inf file."""
    import os
    from .. import _image_size
    if frame_per_color is None:
        frame_per_color = _image_size[0]
    if '.dax' not in dax_filename:
        raise ValueError(f"Wrong input type, .dax file expected for {dax_filename}")
    if not os.path.isfile(dax_filename):
        raise IOError(f"input file:{dax_filename} doesn't exist!")
    _num_frame, _num_color, _dx, _dy = 0, 0, 0, 0
    with open(dax_filename.replace('.dax', '.inf'), 'r') as _info_hd:
        for _line in _info_hd.readlines():
            _line = _line.rstrip()
            if "number of frames" in _line:
                _num_frame = int(_line.split('=')[1])
                _num_color = (_num_frame - 2 * buffer_frame - empty_frame) / frame_per_color
                if _num_color != int(_num_color):
                    raise ValueError("Wrong num_color, should be integer!")
                _num_color = int(_num_color)
            if "frame dimensions" in _line:
                _dx = int(_line.split('=')[1].split('x')[0])
                _dy = int(_line.split('=')[1].split('x')[1])
    return [_num_frame, _dx, _dy], _num_color


def _channel_starts(sel_channels, all_channels, num_buffer_frames, num_empty_frames):
    """First frame of every selected channel (io_tools/load.py:534-540)."""
    _all = [str(_ch) for _ch in all_channels]
    _n = len(_all)
    starts = []
    for _ch in sel_channels:
        if str(_ch) not in _all:
            raise ValueError(f"Wrong input channel:{_ch}, should be within {_all}")
        _i = _all.index(str(_ch))
        starts.append(num_empty_frames + num_buffer_frames + (_i - num_empty_frames - num_buffer_frames) % _n)
    return starts


def split_im_by_channels(im, sel_channels, all_channels, single_im_size=None,
                         num_buffer_frames=10, num_empty_frames=0, skip_frame0=False):
    """io_tools/load.py:524-550.  ``im``: raw movie as ndarray (returns ndarrays) or resident ``DeviceStack``
    (returns resident stacks gathered on the device)."""
    from .. import _image_size
    if single_im_size is None:
        single_im_size = _image_size
    if isinstance(sel_channels, (str, int)):
        sel_channels = [sel_channels]
    if isinstance(all_channels, (str, int)):
        all_channels = [all_channels]
    _n = len(all_channels)
    starts = _channel_starts(sel_channels, all_channels, num_buffer_frames, num_empty_frames)
    if skip_frame0:
        starts = [_s + _n if _s == num_buffer_frames else _s for _s in starts]
    Z = int(single_im_size[0])
    if isinstance(im, L.DeviceStack):
        outs = []
        for _s in starts:
            h = C.c_void_p()
            L.check(L.lib().ia3_stack_deinterleave(im._h, int(_s), int(_n), Z, C.byref(h)))
            outs.append(L.DeviceStack(h, (Z,) + tuple(im.shape[1:]), im.dtype))
        return outs
    return [im[_s:_s + Z * _n:_n].copy() for _s in starts]


class DeviceBuffer(object):
    """A run-constant array (correction profile) resident in HBM."""

    def __init__(self, arr):
        self.arr = np.ascontiguousarray(arr)
        self.dtype_code = 1 if self.arr.dtype == np.float32 else 2
        if self.arr.dtype not in (np.float32, np.float64):
            self.arr = self.arr.astype(np.float64)
            self.dtype_code = 2
        p = C.c_void_p()
        L.check(L.lib().ia3_buffer_upload(L.ptr(self.arr), C.c_size_t(self.arr.nbytes), C.byref(p)))
        self.ptr = p

    def free(self):
        if self.ptr is not None:
            L.lib().ia3_buffer_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _as_buffer(p):
    return p if isinstance(p, DeviceBuffer) else DeviceBuffer(p)


def correct_fov_image(dax_filename, sel_channels,
                      load_file_lock=None,
                      single_im_size=None, all_channels=None,
                      num_buffer_frames=10, num_empty_frames=0,
                      drift=None, calculate_drift=False,
                      drift_channel='488', ref_filename=None,
                      use_autocorr=True, drift_args={},
                      corr_channels=None, correction_folder=None,
                      warp_image=True,
                      hot_pixel_corr=True, hot_pixel_th=4, z_shift_corr=False,
                      illumination_corr=True, illumination_profile=None,
                      bleed_corr=True, bleed_profile=None,
                      chromatic_ref_channel='647', chromatic_corr=True, chromatic_profile=None,
                      gaussian_highpass=False, gauss_sigma=3, gauss_truncate=2,
                      normalization=False, output_dtype=np.uint16,
                      return_drift=False, verbose=True, return_device=False):
    """io_tools/load.py:166-522 — correct one field of view: split channels, hot pixels, z-shift, bleedthrough,
    illumination, (bead drift), warp with drift + chromatic field, (Gaussian high-pass).

    The raw movie is uploaded ONCE as uint16 and every stage runs on resident stacks (hotpix.hip, corrections.hip,
    warp.hip, gauss.hip); only the selected channels come back (or stay resident with ``return_device=True`` — feed
    them to ``ia3_fit_fov_dev`` / ``fit_fov_image``).  ``dax_filename`` may also be the raw (frames, X, Y) uint16
    movie itself.  Profiles: ndarray or ``DeviceBuffer`` (upload them once per run with ``DeviceBuffer``), or None to read them from
    ``correction_folder`` as the reference does (``load_correction_profile``); ``normalization=True`` is outside the
    accelerated path.  With ``warp_image=False`` the images are left unwarped and one spot-translation
    function per selected channel is returned (``chromatic_profile[ch]`` is then the constants dict)."""
    import os
    import time
    from .. import _image_size, _allowed_colors, _corr_channels
    single_im_size = _image_size if single_im_size is None else single_im_size
    all_channels = _allowed_colors if all_channels is None else all_channels
    corr_channels = _corr_channels if corr_channels is None else corr_channels
    if isinstance(dax_filename, np.ndarray):
        _raw_im = dax_filename
    else:
        if not os.path.isfile(dax_filename):
            raise IOError(f"Dax file: {dax_filename} is not a file, exit!")
        if not isinstance(dax_filename, str) or dax_filename[-4:] != '.dax':
            raise IOError(f"Dax file: {dax_filename} has wrong data type, exit!")
        _raw_im = None
    if verbose:
        print(f"- correct the whole fov for image: {dax_filename if _raw_im is None else 'array'}")
        _total_start = time.time()
    if isinstance(sel_channels, (str, int)):
        sel_channels = [str(sel_channels)]
    else:
        sel_channels = [str(ch) for ch in sel_channels]
    single_im_size = np.array(single_im_size, dtype=int)
    all_channels = [str(ch) for ch in all_channels]
    num_buffer_frames, num_empty_frames = int(num_buffer_frames), int(num_empty_frames)
    if drift is None:
        drift = np.zeros(len(single_im_size), dtype=np.float32)
    else:
        drift = np.array(drift, dtype=np.float32)
    if len(drift) != len(single_im_size):
        raise IndexError("drift should have the same dimension as single_im_size.")
    corr_channels = [str(ch) for ch in sorted(corr_channels, key=lambda v: -int(v)) if str(ch) in all_channels]
    _overlap_channels = [_ch for _ch in corr_channels if _ch in sel_channels]
    _load_channels = [_ch for _ch in corr_channels] if (len(_overlap_channels) > 0 and bleed_corr) else []
    for _ch in sel_channels:
        if _ch not in _load_channels:
            _load_channels.append(_ch)
    _drift_channel = str(drift_channel)
    if _drift_channel not in all_channels:
        raise ValueError(f"Wrong input of drift_channel:{_drift_channel}, should be among {all_channels}")
    if calculate_drift and _drift_channel not in _load_channels:
        _load_channels.append(_drift_channel)
    if normalization:
        raise NotImplementedError("normalization=True is outside the accelerated path")
    if np.dtype(output_dtype) != np.uint16:
        raise NotImplementedError("output_dtype other than uint16 is outside the accelerated path")
    # profiles
    if illumination_corr:
        if illumination_profile is None:                                     # :239-246
            illumination_profile = load_correction_profile('illumination', corr_channels=_load_channels,
                                                           correction_folder=correction_folder, all_channels=all_channels,
                                                           ref_channel=chromatic_ref_channel, im_size=single_im_size,
                                                           verbose=verbose)
        if not isinstance(illumination_profile, dict):
            raise TypeError("Wrong input type of illumination_profile, should be dict!")
        for _ch in _load_channels:
            if _ch not in illumination_profile:
                raise KeyError(f"channel:{_ch} not given in illumination_profile")
    _do_bleed = bleed_corr and len(_overlap_channels) > 0
    if _do_bleed:
        if bleed_profile is None:                                            # :255-259
            bleed_profile = load_correction_profile('bleedthrough', corr_channels=corr_channels,
                                                    correction_folder=correction_folder, all_channels=all_channels,
                                                    ref_channel=chromatic_ref_channel, im_size=single_im_size,
                                                    verbose=verbose)
        if not isinstance(bleed_profile, DeviceBuffer):
            bleed_profile = np.array(bleed_profile, dtype=np.float32)
            _nc = len(corr_channels)
            if bleed_profile.shape != (_nc, _nc, single_im_size[-2], single_im_size[-1]):
                raise IndexError(f"Wrong input shape for bleed_profile: {bleed_profile.shape}, should be "
                                 f"{(_nc, _nc, single_im_size[-2], single_im_size[-1])}")
    if chromatic_corr and len(_overlap_channels) > 0:
        if chromatic_profile is None:                                        # :266-281
            chromatic_profile = load_correction_profile('chromatic' if warp_image else 'chromatic_constants',
                                                        corr_channels=corr_channels,
                                                        correction_folder=correction_folder, all_channels=all_channels,
                                                        ref_channel=chromatic_ref_channel, im_size=single_im_size,
                                                        verbose=verbose)
        if not isinstance(chromatic_profile, dict):
            raise TypeError("Wrong input type of chromatic_profile, should be dict!")
        for _ch in _load_channels:
            if _ch in corr_channels and _ch not in chromatic_profile:
                raise KeyError(f"channel:{_ch} not given in chromatic_profile")
    # load + upload once
    if _raw_im is None:
        if load_file_lock is not None:
            load_file_lock.acquire()
        try:
            _raw = load_dax_resident(dax_filename)     # file -> pinned staging -> HBM, no host copy of the movie
        finally:
            if load_file_lock is not None:
                load_file_lock.release()
    else:
        if _raw_im.dtype != np.uint16 or _raw_im.ndim != 3:
            raise TypeError("the raw movie should be a (frames, X, Y) uint16 array")
        _raw = L.DeviceStack.upload(np.ascontiguousarray(_raw_im))
    _num_color = (_raw.shape[0] - 2 * num_buffer_frames - num_empty_frames) / single_im_size[0]
    if _num_color != int(_num_color):
        _raw.free()
        raise ValueError("Wrong num_color, should be integer!")
    _num_color = int(_num_color)
    lib = L.lib()
    _owned = []      # resident stacks to release
    try:
        _ims = split_im_by_channels(_raw, _load_channels, all_channels[:_num_color], single_im_size=single_im_size,
                                    num_buffer_frames=num_buffer_frames, num_empty_frames=num_empty_frames)
        _owned.extend(_ims)
        _raw.free()
        if hot_pixel_corr:                                                   # :323-334
            for _im in _ims:
                nh = C.c_int(0)
                L.check(lib.ia3_remove_hot_pixels_dev(_im._h, C.c_double(0.5), C.c_double(float(hot_pixel_th)), 1,
                                                      C.byref(nh)))
        if z_shift_corr:                                                     # :337-345
            for _im in _ims:
                L.check(lib.ia3_z_shift_correction_dev(_im._h, _im._h))
        if _do_bleed:                                                        # :348-370
            _bp = _as_buffer(bleed_profile)
            _idx = [_load_channels.index(_ch) for _ch in corr_channels]
            _outs = [L.DeviceStack.empty(_ims[_i].shape, np.uint16) for _i in _idx]
            _owned.extend(_outs)
            arr_in = (C.c_void_p * len(_idx))(*[_ims[_i]._h for _i in _idx])
            arr_out = (C.c_void_p * len(_idx))(*[_o._h for _o in _outs])
            L.check(lib.ia3_bleedthrough_correct_dev(arr_in, len(_idx), _bp.ptr, _bp.dtype_code, arr_out))
            for _i, _o in zip(_idx, _outs):
                _ims[_i] = _o
        if illumination_corr:                                                # :373-384
            for _i, _ch in enumerate(_load_channels):
                _ip = _as_buffer(illumination_profile[_ch])
                L.check(lib.ia3_illumination_correct_dev(_ims[_i]._h, _ip.ptr, _ip.dtype_code, _ims[_i]._h))
        if calculate_drift:                                                  # :387-417
            from ..correction_tools.alignment import align_image
            _updated_drift_args = {_k: _v for _k, _v in drift_args.items()}
            _updated_drift_args.update({'all_channels': all_channels, 'ref_all_channels': all_channels,
                                        'drift_channel': drift_channel})
            _drift_corr_args = {'single_im_size': single_im_size, 'num_buffer_frames': num_buffer_frames,
                                'num_empty_frames': num_empty_frames}
            if illumination_corr:
                _drift_corr_args['illumination_profile'] = illumination_profile
            # the bead channel is already resident: align_image crops it on the device (no download / second upload)
            _drift, _drift_flag = align_image(_ims[_load_channels.index(_drift_channel)], ref_filename,
                                              use_autocorr=use_autocorr, correction_args=_drift_corr_args,
                                              verbose=verbose, **_updated_drift_args)
        else:
            _drift = drift.copy()
            _drift_flag = 0
        _chromatic_channels = [_ch for _ch in corr_channels if _ch in sel_channels and _ch != chromatic_ref_channel]
        _warp_functions = []
        if not warp_image:                                                   # :454-486: translate spots instead
            from ..correction_tools.chromatic import generate_chromatic_function
            for _ch in sel_channels:
                if (chromatic_corr and _ch in _chromatic_channels) or _drift.any():
                    if chromatic_corr and _ch in _chromatic_channels:
                        _warp_functions.append(generate_chromatic_function(chromatic_profile[_ch], _drift))
                    else:
                        _warp_functions.append(generate_chromatic_function(None, _drift))
                else:
                    _warp_functions.append(lambda _spots: _spots)
        for _ch in (sel_channels if warp_image else []):                     # :424-453
            # the reference's resampling block sits INSIDE its `if verbose:` (:436-453 are indented under the
            # print of :435), so a silent call returns unwarped images; kept, so outputs match call for call
            if ((chromatic_corr and _ch in _chromatic_channels) or _drift.any()) and verbose:
                _i = _load_channels.index(_ch)
                _field, _fdt = None, 0
                if chromatic_corr and _ch in _chromatic_channels and chromatic_profile[_ch] is not None:
                    _fb = _as_buffer(chromatic_profile[_ch])
                    if _fb.arr.shape != (3,) + tuple(_ims[_i].shape):
                        raise IndexError(f"chromatic_profile[{_ch}] shape {_fb.arr.shape} should be "
                                         f"{(3,) + tuple(_ims[_i].shape)}")
                    _field, _fdt = _fb.ptr, _fb.dtype_code
                _d = np.ascontiguousarray(_drift if _drift.any() else np.zeros(3), dtype=np.float64)
                _out = L.DeviceStack.empty(_ims[_i].shape, np.uint16)
                _owned.append(_out)
                L.check(lib.ia3_warp3d_dev(_ims[_i]._h, L.dptr(_d), _field, _fdt, 3, L.MODE_NEAREST,
                                           C.c_double(0.0), _out._h))
                _ims[_i] = _out
        if gaussian_highpass:                                                # :489-498 (every loaded channel)
            w, r = L.gaussian_taps(gauss_sigma, gauss_truncate)
            for _i, _im in enumerate(_ims):
                _out = L.DeviceStack.empty(_im.shape, np.uint16)
                _owned.append(_out)
                L.check(lib.ia3_gaussian_highpass_dev(_im._h, C.c_double(gauss_sigma), C.c_double(gauss_truncate),
                                                      L.dptr(w), int(r), _out._h))
                _ims[_i] = _out
        _sel = [_ims[_load_channels.index(_ch)] for _ch in sel_channels]
        if return_device:
            _sel_ims = _sel
            _owned = [_o for _o in _owned if not any(_o is _s for _s in _sel)]
        else:
            _sel_ims = [_s.download() for _s in _sel]
    finally:
        for _o in _owned:
            _o.free()
        _raw.free()
    if verbose:
        print(f"-- finish correction in {time.time()-_total_start:.3f}s")
    _return_args = [_sel_ims]
    if not warp_image:
        _return_args.append(_warp_functions)
    if return_drift:
        _return_args.extend([_drift, _drift_flag])
    return tuple(_return_args)


# ---------------------------------------------------------------------------------------------------------------
# many movies through one pipelined call (ia3_process_movies): correct_fov_image + fit_fov_image per selected channel
# ---------------------------------------------------------------------------------------------------------------

class MoviePlan(object):
    """Everything that is the same for every movie of a run — channel layout, correction profiles and the reference bead
    image resident in HBM, seeding / fitting parameters — as the ``ia3_movie_params`` that ``ia3_process_movies`` takes.

    Arguments are those of ``correct_fov_image`` (io_tools/load.py:166-522) plus, as in
    ``classes/batch_functions.py:60-302``: ``ref_image`` (the corrected reference bead image: ndarray, resident
    ``DeviceStack`` or a .dax path that ``align_image`` would load), ``seed_th`` (threshold per selected channel),
    ``fitting_args`` (the keyword arguments the caller would hand to ``fit_fov_image``) and ``fit_spots``.
    ``run(movies, ...)`` then processes raw movies (ndarrays or .dax paths): uploads, corrections / drift / warps and group
    fits of different movies overlap on library threads.  Results are those of the per-movie calls.

    Raises ``NotImplementedError`` for options the pipelined entry does not cover (the caller then goes movie by movie):
    bead-fitting drift (``use_autocorr=False``), ``warp_image=False``, ``normalization``, a non-uint16 output type, seeds /
    seed masks / percentile thresholds in ``fitting_args``."""

    def __init__(self, sel_channels, ref_image=None, single_im_size=None, all_channels=None,
                 num_buffer_frames=10, num_empty_frames=0, calculate_drift=True, drift_channel='488',
                 use_autocorr=True, drift_args={}, corr_channels=None, correction_folder=None, warp_image=True,
                 hot_pixel_corr=True, hot_pixel_th=4, z_shift_corr=False,
                 illumination_corr=True, illumination_profile=None, bleed_corr=True, bleed_profile=None,
                 chromatic_ref_channel='647', chromatic_corr=True, chromatic_profile=None,
                 gaussian_highpass=False, gauss_sigma=3, gauss_truncate=2, normalization=False, output_dtype=np.uint16,
                 verbose=True, seed_th=None, fitting_args={}, fit_spots=True, frames=None,
                 correct_threads=2, fit_group_images=12, upload_ahead=2):
        from .. import _image_size, _allowed_colors, _corr_channels
        from ..correction_tools.alignment import generate_drift_crops, DEFAULT_NORMALIZATION
        single_im_size = _image_size if single_im_size is None else single_im_size
        all_channels = _allowed_colors if all_channels is None else all_channels
        corr_channels = _corr_channels if corr_channels is None else corr_channels
        if isinstance(sel_channels, (str, int)):
            sel_channels = [str(sel_channels)]
        sel_channels = [str(ch) for ch in sel_channels]
        single_im_size = np.array(single_im_size, dtype=int)
        all_channels = [str(ch) for ch in all_channels]
        num_buffer_frames, num_empty_frames = int(num_buffer_frames), int(num_empty_frames)
        if normalization:
            raise NotImplementedError("normalization=True is outside the accelerated path")
        if np.dtype(output_dtype) != np.uint16:
            raise NotImplementedError("output_dtype other than uint16 is outside the accelerated path")
        if not warp_image:
            raise NotImplementedError("warp_image=False (spot translation functions) goes movie by movie")
        if calculate_drift and not use_autocorr:
            raise NotImplementedError("bead-fitting drift (use_autocorr=False) goes movie by movie")
        # channel bookkeeping exactly as correct_fov_image (:196-236)
        corr_channels = [str(ch) for ch in sorted(corr_channels, key=lambda v: -int(v)) if str(ch) in all_channels]
        _overlap_channels = [_ch for _ch in corr_channels if _ch in sel_channels]
        _load_channels = [_ch for _ch in corr_channels] if (len(_overlap_channels) > 0 and bleed_corr) else []
        for _ch in sel_channels:
            if _ch not in _load_channels:
                _load_channels.append(_ch)
        _drift_channel = str(drift_channel)
        if _drift_channel not in all_channels:
            raise ValueError(f"Wrong input of drift_channel:{_drift_channel}, should be among {all_channels}")
        if calculate_drift and _drift_channel not in _load_channels:
            _load_channels.append(_drift_channel)
        if len(_load_channels) > L.MOVIE_MAXCH:
            raise NotImplementedError("more than %d channels per movie" % L.MOVIE_MAXCH)
        Z, X, Y = (int(v) for v in single_im_size)
        self.sel_channels, self.load_channels, self.shape = sel_channels, _load_channels, (Z, X, Y)
        self._keep = []
        p = L.MovieParams()
        if frames is None:
            frames = Z * len(all_channels) + 2 * num_buffer_frames + num_empty_frames
        _num_color = (int(frames) - 2 * num_buffer_frames - num_empty_frames) / Z
        if _num_color != int(_num_color):
            raise ValueError("Wrong num_color, should be integer!")
        _num_color = int(_num_color)
        p.frames, p.X, p.Y, p.Z = int(frames), X, Y, Z
        starts = _channel_starts(_load_channels, all_channels[:_num_color], num_buffer_frames, num_empty_frames)
        p.n_load, p.load_step = len(_load_channels), _num_color
        for _i, _s in enumerate(starts):
            p.load_start[_i] = int(_s)
        p.n_sel = len(sel_channels)
        for _i, _ch in enumerate(sel_channels):
            p.sel[_i] = _load_channels.index(_ch)
        p.hot_pixel_corr, p.hot_pixel_th = (1 if hot_pixel_corr else 0), float(hot_pixel_th)
        p.z_shift_corr = 1 if z_shift_corr else 0
        # profiles (uploaded once, or handed over as DeviceBuffers)
        if illumination_corr:
            if illumination_profile is None:
                illumination_profile = load_correction_profile('illumination', corr_channels=_load_channels,
                                                               correction_folder=correction_folder, all_channels=all_channels,
                                                               ref_channel=chromatic_ref_channel, im_size=single_im_size,
                                                               verbose=False)
            if not isinstance(illumination_profile, dict):
                raise TypeError("Wrong input type of illumination_profile, should be dict!")
            for _i, _ch in enumerate(_load_channels):
                if _ch not in illumination_profile:
                    raise KeyError(f"channel:{_ch} not given in illumination_profile")
                _ip = _as_buffer(illumination_profile[_ch])
                if tuple(_ip.arr.shape) != (X, Y):
                    raise IndexError(f"illumination profile shape {_ip.arr.shape} should be {(X, Y)}")
                self._keep.append(_ip)
                p.illum_profile[_i], p.illum_dtype[_i] = _ip.ptr.value, _ip.dtype_code
        if bleed_corr and len(_overlap_channels) > 0:
            if bleed_profile is None:
                bleed_profile = load_correction_profile('bleedthrough', corr_channels=corr_channels,
                                                        correction_folder=correction_folder, all_channels=all_channels,
                                                        ref_channel=chromatic_ref_channel, im_size=single_im_size,
                                                        verbose=False)
            if not isinstance(bleed_profile, DeviceBuffer):
                bleed_profile = np.array(bleed_profile, dtype=np.float32)
            _bp = _as_buffer(bleed_profile)
            _nc = len(corr_channels)
            if tuple(_bp.arr.shape) != (_nc, _nc, X, Y):
                raise IndexError(f"Wrong input shape for bleed_profile: {_bp.arr.shape}, should be {(_nc, _nc, X, Y)}")
            self._keep.append(_bp)
            p.n_bleed = _nc
            for _i, _ch in enumerate(corr_channels):
                p.bleed_idx[_i] = _load_channels.index(_ch)
            p.bleed_profile, p.bleed_dtype = _bp.ptr.value, _bp.dtype_code
        _chromatic_channels = [_ch for _ch in corr_channels if _ch in sel_channels and _ch != chromatic_ref_channel]
        if chromatic_corr and len(_overlap_channels) > 0:
            if chromatic_profile is None:
                chromatic_profile = load_correction_profile('chromatic', corr_channels=corr_channels,
                                                            correction_folder=correction_folder, all_channels=all_channels,
                                                            ref_channel=chromatic_ref_channel, im_size=single_im_size,
                                                            verbose=False)
            if not isinstance(chromatic_profile, dict):
                raise TypeError("Wrong input type of chromatic_profile, should be dict!")
            for _ch in _load_channels:
                if _ch in corr_channels and _ch not in chromatic_profile:
                    raise KeyError(f"channel:{_ch} not given in chromatic_profile")
        # the reference resamples inside `if verbose:` (:434-453): a silent call returns unwarped images
        p.warp = 1 if verbose else 0
        for _i, _ch in enumerate(sel_channels):
            _chrom = bool(chromatic_corr and _ch in _chromatic_channels)
            p.warp_always[_i] = 1 if _chrom else 0
            if _chrom and chromatic_profile[_ch] is not None:
                _fb = _as_buffer(chromatic_profile[_ch])
                if tuple(_fb.arr.shape) != (3, Z, X, Y):
                    raise IndexError(f"chromatic_profile[{_ch}] shape {_fb.arr.shape} should be {(3, Z, X, Y)}")
                self._keep.append(_fb)
                p.chrom_field[_i], p.chrom_dtype[_i] = _fb.ptr.value, _fb.dtype_code
        if gaussian_highpass:
            p.highpass_sigma, p.highpass_truncate = float(gauss_sigma), float(gauss_truncate)
        # drift
        p.drift_idx = -1
        self.measure_drift = bool(calculate_drift)
        if calculate_drift:
            p.drift_idx = _load_channels.index(_drift_channel)
            _da = dict(drift_args)
            for _k in ('all_channels', 'ref_all_channels', 'drift_channel', 'verbose', 'detailed_verbose', 'use_autocorr',
                       'correction_args', 'match_distance_th', 'fitting_args'):
                _da.pop(_k, None)
            crop_list = _da.pop('crop_list', None)
            if crop_list is None:
                crop_list = generate_drift_crops(single_im_size)
            crop_list = np.array(crop_list, dtype=int)
            if crop_list.ndim != 3 or crop_list.shape[1:] != (3, 2) or len(crop_list) > 8:
                raise IndexError("crop should be 3x2 np.ndarray.")
            crop_list[:, :, 0] = np.maximum(crop_list[:, :, 0], 0)
            crop_list[:, :, 1] = np.minimum(crop_list[:, :, 1], np.array([Z, X, Y])[None, :])
            p.n_crops = len(crop_list)
            for _i, _v in enumerate(crop_list.reshape(-1)):
                p.crops[_i] = int(_v)
            p.precision_fold = int(_da.pop('precision_fold', 100))
            p.min_good_drifts = int(_da.pop('min_good_drifts', 3))
            p.drift_diff_th = float(_da.pop('drift_diff_th', 1.))
            p.normalization = 1 if DEFAULT_NORMALIZATION == "phase" else 0
            if _da:
                raise NotImplementedError("drift_args %s go movie by movie" % sorted(_da))
            if isinstance(ref_image, str):
                from ..correction_tools.alignment import align_image   # noqa: F401  (the loader it uses)
                _corr = {'single_im_size': single_im_size, 'num_buffer_frames': num_buffer_frames,
                         'num_empty_frames': num_empty_frames, 'hot_pixel_corr': True, 'z_shift_corr': False,
                         'bleed_corr': False, 'chromatic_corr': False, 'normalization': False,
                         'illumination_corr': bool(illumination_corr)}
                if illumination_corr:
                    _corr['illumination_profile'] = illumination_profile
                ref_image = correct_fov_image(ref_image, [_drift_channel], all_channels=all_channels, calculate_drift=False,
                                              return_drift=False, verbose=False, return_device=True, **_corr)[0][0]
                self._keep.append(ref_image)
            if isinstance(ref_image, np.ndarray):
                if tuple(ref_image.shape) != (Z, X, Y):
                    raise IndexError(f"shape of reference image:{ref_image.shape} should be {(Z, X, Y)}")
                ref_image = L.DeviceStack.upload(ref_image if ref_image.dtype == np.uint16 else ref_image.astype(np.uint16))
                self._keep.append(ref_image)
            if not isinstance(ref_image, L.DeviceStack):
                raise TypeError(f"ref_filename should be np.ndarray or string of path, but {type(ref_image)} is given")
            if ref_image.dtype != np.uint16 or tuple(ref_image.shape) != (Z, X, Y):
                raise TypeError("the resident reference bead stack must be uint16 of the image size")
            self._keep.append(ref_image)
            p.ref_bead = ref_image._h
            # the reference crops' spectra, once for every run() of this plan
            from ..correction_tools.alignment import DriftReference
            self.drift_reference = DriftReference(ref_image, crop_list)
            p.drift_ref = self.drift_reference._h
        # seeding + fitting (spot_tools/fitting.py:169-262 through classes/batch_functions.py:248-300)
        p.fit_spots = 1 if fit_spots else 0
        fa = dict(fitting_args)
        for _k in ('seeds', 'seed_mask'):
            if fa.get(_k, None) is not None:
                raise NotImplementedError("fitting_args['%s'] goes movie by movie" % _k)
            fa.pop(_k, None)
        if fa.pop('use_percentile', False):
            raise NotImplementedError("percentile thresholds go movie by movie")
        fa.pop('th_seed_per', None)
        fa.pop('th_seed', None)      # set per channel below, as the reference's loop does
        fa.pop('verbose', None)
        if not fa.pop('remove_boundary_points', True):
            raise NotImplementedError("remove_boundary_points=False goes movie by movie")
        seeding_kwargs = dict(fa.pop('seeding_kwargs', {}))
        for _k in ('sel_center', 'use_percentile'):
            if seeding_kwargs.pop(_k, None):
                raise NotImplementedError("seeding_kwargs['%s'] goes movie by movie" % _k)
        for _k in ('seed_radius', 'th_seed_per', 'return_h', 'verbose'):
            seeding_kwargs.pop(_k, None)
        seed_kw = dict(max_num_seeds=fa.pop('max_num_seeds', 500), use_dynamic_th=fa.pop('use_dynamic_th', True),
                       dynamic_niters=fa.pop('dynamic_niters', 10), min_dynamic_seeds=fa.pop('min_dynamic_seeds', 1),
                       remove_hot_pixel=fa.pop('remove_hot_pixel', True), **seeding_kwargs)
        if seed_th is None:
            seed_th = {}
        for _i, _ch in enumerate(sel_channels):
            _th = float(seed_th[_ch] if isinstance(seed_th, dict) else seed_th[_i]) if fit_spots else 300.
            _sp, _k = L.make_seed_params(_th, **seed_kw)
            self._keep.append(_k)
            p.seed[_i] = _sp
        fit_radius = int(fa.pop('fit_radius', 5))
        p.fit = L.make_fit_params(radius_fit=fit_radius, **dict(fa.pop('fitting_args', {})))
        normalize_background, normalize_local = fa.pop('normalize_background', False), fa.pop('normalize_local', False)
        background_args = dict(fa.pop('background_args', {}))
        if normalize_local or normalize_background:
            background_args.pop('make_plot', None)
            _dt = background_args.pop('dtype', _image_dtype)
            edges = _background_edges(_dt if _dt is not None else np.uint16, background_args.pop('bin_size', 10))
            self._keep.append(edges)
            p.normalize = 2 if normalize_local else 1
            p.bg_crop_size = fit_radius * 2
            p.bg_edges, p.bg_n_edges = L.dptr(edges), len(edges)
            p.bg_max_iter = int(background_args.pop('max_iter', 10))
            if background_args:
                raise NotImplementedError("background_args %s go movie by movie" % sorted(background_args))
        if fa:
            raise NotImplementedError("fitting_args %s go movie by movie" % sorted(fa))
        p.correct_threads, p.fit_group_images, p.upload_ahead = int(correct_threads), int(fit_group_images), int(upload_ahead)
        self.params = p

    def run(self, movies, drifts_in=None, measure_drift=None, want_images=False, capacity=16384):
        """``movies``: raw (frames, X, Y) uint16 arrays or .dax paths.  Per movie a dict: ``tables`` (one (M,11) float32
        table per selected channel), ``drift``, ``drift_flag``, ``images`` (with ``want_images``), ``n_seeds``, ``n_iter``,
        ``ms``.  ``drifts_in`` / ``measure_drift`` (bool or list): a known drift per movie instead of measuring it."""
        items = []
        for m in movies:
            if isinstance(m, str):
                n, w, h, big = _dax_info(m)
                if (n, w, h) != (int(self.params.frames), self.shape[1], self.shape[2]):
                    raise ValueError(f"{m}: movie of {(n, w, h)} does not have the planned layout "
                                     f"{(int(self.params.frames), self.shape[1], self.shape[2])}")
                items.append((m, 0, big))
            else:
                items.append(m)
        if measure_drift is None:
            measure_drift = self.measure_drift
        if drifts_in is not None:   # the chain carries a given drift as float32 (io_tools/load.py:202-206)
            drifts_in = [None if d is None else np.array(d, dtype=np.float32).astype(np.float64) for d in drifts_in]
        return L.process_movies(self.params, items, drifts_in=drifts_in, measure_drift=measure_drift,
                                want_images=want_images, capacity=capacity)
