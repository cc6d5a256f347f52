"""TEST HARNESS — not part of the product package.

A step-by-step driver with the reference's ``DaxProcesser`` interface (classes/preprocess.py:337-1260) over the
library's device operators.  The reference keeps its own class (it is a caller of the hot path); this
copy exists so that the end-to-end parity fixtures — the reference class run on a synthetic .dax movie,
oracle/make_golden.py::daxp_golden — can be replayed against the same sequence of operator calls.
"""
import numpy as np
from imageanalysis3_amd import _image_size  # noqa: F401
from imageanalysis3_amd.classes.preprocess import Spots3D, ImageCrop, ImageCrop_3d, _3d_spot_infos  # noqa: F401

# ---------------------------------------------------------------------------------------------------------------
# DaxProcesser (reference: classes/preprocess.py:337-1260) — the step API over one .dax movie
# ---------------------------------------------------------------------------------------------------------------
default_im_size = np.array([50, 2048, 2048])
default_pixel_sizes = np.array([250, 108, 108])
default_channels = ['750', '647', '561', '488', '405']
default_ref_channel = '647'
default_dapi_channel = '405'
default_num_buffer_frames = 0
default_num_empty_frames = 0
default_seed_th = 1000


class DaxProcesser():
    """classes/preprocess.py:337-1260 — load one .dax movie, correct it step by step, fit spots.

    Same constructor, step methods, attribute names (``im_<ch>``, ``spots_<ch>``, ``drift``, ``correction_log`` ...) and
    return conventions as the reference class.  The channel images live in HBM between steps (``im_<ch>`` is
    downloaded when it is read and uploaded when it is assigned), every step is the device kernel of the operator the
    reference calls there, with this class's own arithmetic where it differs from ``correct_fov_image`` (uint16 hot
    pixel votes, float64 bleedthrough accumulation and min-max rescale, rescaled illumination, drift applied before
    the chromatic field).  Profiles must be passed in (``correction_pf=...``): reading the pickled correction folder,
    HDF5 output and segmentation-driven fitting are outside the accelerated path."""

    def __init__(self, ImageFilename, CorrectionFolder=None, Channels=None, DriftChannel=None, DapiChannel=None,
                 verbose=True):
        import os
        object.__setattr__(self, "_dev", {})        # channel -> DeviceStack
        object.__setattr__(self, "_host", {})       # channel -> downloaded copy (valid until the stack changes)
        if isinstance(ImageFilename, str) and os.path.isfile(ImageFilename) \
                and ImageFilename.split(os.extsep)[-1] == 'dax':
            self.filename = ImageFilename
        elif not isinstance(ImageFilename, str):
            raise TypeError(f"Wrong input type ({type(ImageFilename)}) for ImageFilename.")
        elif ImageFilename.split(os.extsep)[-1] != 'dax':
            raise TypeError("Wrong input file extension, should be .dax")
        else:
            raise OSError(f"image file: {ImageFilename} doesn't exist, exit.")
        if verbose:
            print(f"Initialize DaxProcesser for file:{ImageFilename}")
        self.inf_filename = self.filename.replace('.dax', '.inf')
        self.off_filename = self.filename.replace('.dax', '.off')
        self.power_filename = self.filename.replace('.dax', '.power')
        self.xml_filename = self.filename.replace('.dax', '.xml')
        self.correction_folder = CorrectionFolder
        if Channels is None:
            _loaded_channels = DaxProcesser._FindDaxChannels(self.filename, verbose=verbose)
            self.channels = default_channels if _loaded_channels is None else _loaded_channels
        elif isinstance(Channels, (list, np.ndarray)):
            self.channels = list(Channels)
        else:
            raise TypeError("Wrong input type for Channels")
        if DriftChannel is not None and str(DriftChannel) in self.channels:
            setattr(self, 'drift_channel', str(DriftChannel))
        if DapiChannel is not None and str(DapiChannel) in self.channels:
            setattr(self, 'dapi_channel', str(DapiChannel))
        self.correction_log = {_ch: {} for _ch in self.channels}
        self.verbose = verbose

    # -- im_<ch> attributes are views of the resident stacks -------------------------------------------------------
    def __getattr__(self, name):
        if name.startswith("im_"):
            _dev = object.__getattribute__(self, "_dev")
            ch = name[3:]
            if ch in _dev:
                _host = object.__getattribute__(self, "_host")
                if ch not in _host:
                    _host[ch] = _dev[ch].download()
                return _host[ch]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        from imageanalysis3_amd import _lib as L
        if name.startswith("im_") and isinstance(value, (np.ndarray, L.DeviceStack)):
            self._set_stack(name[3:], value if isinstance(value, L.DeviceStack) else L.DeviceStack.upload(L.as_stack_array(value)))
        else:
            object.__setattr__(self, name, value)

    def __delattr__(self, name):
        if name.startswith("im_") and name[3:] in self._dev:
            self._dev.pop(name[3:]).free()
            self._host.pop(name[3:], None)
        else:
            object.__delattr__(self, name)

    def _set_stack(self, ch, stack):
        old = self._dev.get(ch)
        if old is not None and old is not stack:
            old.free()
        self._dev[ch] = stack
        self._host.pop(ch, None)

    def device_image(self, ch):
        """The resident stack of a channel (``DeviceStack``), e.g. to hand to ``fit_fov_image`` / ``align_image``."""
        return self._dev[str(ch)]

    def __del__(self):
        try:
            for _s in self._dev.values():
                _s.free()
        except Exception:
            pass

    def _check_existance(self):
        import os
        return os.path.isfile(self.filename) and os.path.isfile(self.inf_filename) and os.path.isfile(self.off_filename) \
            and os.path.isfile(self.power_filename) and os.path.isfile(self.xml_filename)

    # -- steps ------------------------------------------------------------------------------------------------------
    def _load_image(self, sel_channels=None, ImSize=None, NbufferFrame=default_num_buffer_frames,
                    NemptyFrame=default_num_empty_frames, save_attrs=True, overwrite=False):
        """:400-462 — read the movie, gather the selected channels on the device."""
        import time
        from imageanalysis3_amd import _lib as L
        from imageanalysis3_amd.io_tools.load import load_dax_resident, split_im_by_channels
        _load_start = time.time()
        if not hasattr(self, 'loaded_channels'):
            setattr(self, 'loaded_channels', [])
        if sel_channels is None:
            _sel_channels = self.channels
        elif isinstance(sel_channels, list):
            _sel_channels = [str(_ch) for _ch in sel_channels]
        elif isinstance(sel_channels, (str, int)):
            _sel_channels = [str(sel_channels)]
        else:
            raise ValueError("Invalid input for sel_channels")
        _loading_channels = [_ch for _ch in sorted(_sel_channels, key=lambda v: self.channels.index(v))
                             if not (hasattr(self, f"im_{_ch}") and not overwrite)]
        if ImSize is None:
            self.image_size = DaxProcesser._FindImageSize(self.filename, channels=self.channels, NbufferFrame=NbufferFrame,
                                                          verbose=self.verbose)
        else:
            self.image_size = np.array(ImSize, dtype=np.int32)
        _raw = load_dax_resident(self.filename)
        try:
            _ims = split_im_by_channels(_raw, _loading_channels, all_channels=self.channels, single_im_size=self.image_size,
                                        num_buffer_frames=NbufferFrame, num_empty_frames=NemptyFrame)
        finally:
            _raw.free()
        if self.verbose:
            print(f"- Loaded images for channels:{_loading_channels} in {time.time()-_load_start:.3f}s.")
        if save_attrs:
            for _ch, _im in zip(_loading_channels, _ims):
                self._set_stack(_ch, _im)
            setattr(self, 'num_buffer_frames', NbufferFrame)
            setattr(self, 'num_empty_frames', NemptyFrame)
            self.loaded_channels.extend(_loading_channels)
            self.loaded_channels = [_ch for _ch in sorted(self.loaded_channels, key=lambda v: self.channels.index(v))]
            return
        _out = [_im.download() for _im in _ims]
        for _im in _ims:
            _im.free()
        return _out, _loading_channels

    def _finish(self, _chs, _stacks, save_attrs, log_key):
        """Common tail of the correction steps: keep the new stacks (and log) or hand back host copies."""
        if save_attrs:
            for _ch, _st in zip(_chs, _stacks):
                if _st is not None:
                    self._set_stack(_ch, _st)
                    if log_key:
                        self.correction_log[_ch][log_key] = True
            return None
        _out = []
        for _st in _stacks:
            _out.append(None if _st is None else _st.download())
            if _st is not None:
                _st.free()
        return _out, _chs

    def _corr_bleedthrough(self, correction_channels=None, correction_pf=None, correction_folder=None, rescale=True,
                           save_attrs=True, overwrite=False):
        """:464-541."""
        import ctypes as C
        from imageanalysis3_amd import _lib as L
        from imageanalysis3_amd.io_tools.load import _as_buffer
        if correction_channels is None:
            correction_channels = self.loaded_channels
        _correction_channels = [str(_ch) for _ch in correction_channels
                                if str(_ch) != getattr(self, 'drift_channel', None)
                                and str(_ch) != getattr(self, 'dapi_channel', None)]
        _logs = [self.correction_log[_ch].get('corr_bleedthrough', False) for _ch in _correction_channels]
        if np.array(_logs).all():
            if self.verbose:
                print("- Correct bleedthrough already finished, skip. ")
            return
        if correction_folder is None:
            correction_folder = self.correction_folder
        if correction_pf is None:                                            # :494-502
            from imageanalysis3_amd.io_tools.load import load_correction_profile
            correction_pf = load_correction_profile('bleedthrough', _correction_channels,
                                                    correction_folder=correction_folder,
                                                    ref_channel=_correction_channels[0], all_channels=self.channels,
                                                    im_size=self.image_size, verbose=self.verbose)
        if any(_ch not in self._dev for _ch in _correction_channels):
            raise NotImplementedError("bleedthrough correction needs every correction channel loaded")
        _n = len(_correction_channels)
        _pf = _as_buffer(correction_pf)
        if _pf.arr.shape != (_n, _n) + tuple(self.image_size[1:]):
            raise IndexError(f"correction_pf shape {_pf.arr.shape} should be {(_n, _n) + tuple(self.image_size[1:])}")
        _ins = [self._dev[_ch] for _ch in _correction_channels]
        _outs = [L.DeviceStack.empty(_i.shape, np.uint16) for _i in _ins]
        arr_in = (C.c_void_p * _n)(*[_i._h for _i in _ins])
        arr_out = (C.c_void_p * _n)(*[_o._h for _o in _outs])
        L.check(L.lib().ia3_bleedthrough_rescale_dev(arr_in, _n, _pf.ptr, _pf.dtype_code, int(bool(rescale)), arr_out))
        return self._finish(_correction_channels, _outs, save_attrs, 'corr_bleedthrough')

    def _corr_hot_pixels_3D(self, correction_channels=None, hot_pixel_th: float = 0.5, hot_pixel_num_th: float = 4,
                            save_attrs: bool = True):
        """:543-603 — ``correction_tools.filter.Remove_Hot_Pixels`` in the image dtype (uint16 sums wrap)."""
        import ctypes as C
        from imageanalysis3_amd import _lib as L
        if correction_channels is None:
            correction_channels = self.loaded_channels
        _correction_channels = [str(_ch) for _ch in correction_channels]
        _logs = [self.correction_log[_ch].get('corr_hot_pixel', False) for _ch in _correction_channels]
        if np.array(_logs).all():
            if self.verbose:
                print("- Correct hot_pixel already finished, skip. ")
            return
        _correction_channels = [_ch for _ch, _log in zip(_correction_channels, _logs) if not _log]
        _stacks = []
        for _ch in _correction_channels:
            if _ch not in self._dev:
                _stacks.append(None)
                continue
            _st = self._dev[_ch] if save_attrs else self._dev[_ch].crop(np.array([[0, s] for s in self._dev[_ch].shape]))
            nh = C.c_int(0)
            L.check(L.lib().ia3_remove_hot_pixels_dev(_st._h, C.c_double(float(hot_pixel_th)),
                                                      C.c_double(float(hot_pixel_num_th)), 0, C.byref(nh)))
            if save_attrs:
                self._host.pop(_ch, None)
                self.correction_log[_ch]['corr_hot_pixel'] = True
            else:
                _stacks.append(_st)
        if save_attrs:
            return
        return self._finish(_correction_channels, _stacks, False, None)

    def _corr_illumination(self, correction_channels=None, correction_pf=None, correction_folder=None, rescale=True,
                           save_attrs=True, overwrite=False):
        """:605-680."""
        from imageanalysis3_amd import _lib as L
        from imageanalysis3_amd.io_tools.load import _as_buffer
        if correction_channels is None:
            correction_channels = self.loaded_channels
        _correction_channels = [str(_ch) for _ch in correction_channels]
        _logs = [self.correction_log[_ch].get('corr_illumination', False) for _ch in _correction_channels]
        if np.array(_logs).all():
            if self.verbose:
                print("- Correct illumination already finished, skip. ")
            return
        _correction_channels = [_ch for _ch, _log in zip(_correction_channels, _logs) if not _log]
        if correction_folder is None:
            correction_folder = self.correction_folder
        if correction_pf is None:                                            # :635-643
            from imageanalysis3_amd.io_tools.load import load_correction_profile
            correction_pf = load_correction_profile('illumination', _correction_channels,
                                                    correction_folder=correction_folder,
                                                    ref_channel=_correction_channels[0], all_channels=self.channels,
                                                    im_size=self.image_size, verbose=self.verbose)
        _stacks = []
        for _ch in _correction_channels:
            if _ch not in self._dev:
                _stacks.append(None)
                continue
            _pf = _as_buffer(correction_pf[_ch])
            _out = L.DeviceStack.empty(self._dev[_ch].shape, np.uint16)
            L.check(L.lib().ia3_illumination_rescale_dev(self._dev[_ch]._h, _pf.ptr, _pf.dtype_code, int(bool(rescale)), _out._h))
            _stacks.append(_out)
        return self._finish(_correction_channels, _stacks, save_attrs, 'corr_illumination')

    def _corr_chromatic_functions(self, correction_channels=None, correction_pf=None, correction_folder=None,
                                  ref_channel=default_ref_channel, save_attrs=True, overwrite=False):
        """:682-749 — spot-translation functions instead of warping."""
        from imageanalysis3_amd.correction_tools.chromatic import generate_chromatic_function
        if correction_channels is None:
            correction_channels = self.loaded_channels
        _correction_channels = [str(_ch) for _ch in correction_channels
                                if str(_ch) != getattr(self, 'drift_channel', None)
                                and str(_ch) != getattr(self, 'dapi_channel', None)]
        _logs = [self.correction_log[_ch].get('corr_chromatic', False)
                 or self.correction_log[_ch].get('corr_chromatic_function', False) for _ch in _correction_channels]
        if np.array(_logs).all():
            return
        _correction_channels = [_ch for _ch, _log in zip(_correction_channels, _logs) if not _log]
        if correction_folder is None:
            correction_folder = self.correction_folder
        if correction_pf is None:                                            # :716-724
            from imageanalysis3_amd.io_tools.load import load_correction_profile
            correction_pf = load_correction_profile('chromatic_constants', _correction_channels,
                                                    correction_folder=correction_folder, all_channels=self.channels,
                                                    ref_channel=ref_channel, im_size=self.image_size,
                                                    verbose=self.verbose)
        _drift = getattr(self, 'drift', np.zeros(len(self.image_size)))
        _funcs = []
        for _ch in _correction_channels:
            _func = generate_chromatic_function(correction_pf[_ch], _drift)
            if save_attrs:
                setattr(self, f"chromatic_func_{_ch}", _func)
            else:
                _funcs.append(_func)
        if save_attrs:
            for _ch in _correction_channels:
                self.correction_log[_ch]['corr_chromatic_function'] = True
            return
        return _funcs

    def _calculate_drift(self, RefImage, DriftChannel='488', precise_align=True, use_autocorr=True, drift_kwargs={},
                         save_attr=True, save_ref_im=False, overwrite=False):
        """:751-848."""
        import os
        from imageanalysis3_amd import _lib as L
        from imageanalysis3_amd.correction_tools.alignment import align_image, phase_cross_correlation
        if hasattr(self, 'drift') and hasattr(self, 'drift_flag') and not overwrite:
            return self.drift, self.drift_flag
        if DriftChannel is None and hasattr(self, 'drift_channel'):
            DriftChannel = getattr(self, 'drift_channel')
        elif DriftChannel is not None:
            DriftChannel = str(DriftChannel)
        else:
            raise ValueError(f"Wrong input value for DriftChannel: {DriftChannel}")
        _own = None
        if DriftChannel in self.channels and DriftChannel in self._dev:
            _DriftImage = self._dev[DriftChannel]
        elif DriftChannel in self.channels:
            _ims, _ = self._load_image(sel_channels=[DriftChannel], ImSize=self.image_size,
                                       NbufferFrame=self.num_buffer_frames, NemptyFrame=self.num_empty_frames,
                                       save_attrs=False)
            _DriftImage = _own = L.DeviceStack.upload(_ims[0])
        else:
            raise AttributeError(f"DriftChannel:{DriftChannel} image doesn't exist, exit.")
        _own_ref = None
        try:
            if isinstance(RefImage, str) and os.path.isfile(RefImage):
                if RefImage == self.filename:
                    _drift, _drift_flag = np.zeros(len(self.image_size)), 0
                    if save_attr:
                        setattr(self, 'drift_channel', DriftChannel)
                        if save_ref_im:
                            setattr(self, 'ref_im', getattr(self, f"im_{DriftChannel}"))
                        setattr(self, 'drift', _drift)
                        setattr(self, 'drift_flag', _drift_flag)
                        return
                    return _drift, _drift_flag
                _ref_cls = DaxProcesser(RefImage, CorrectionFolder=self.correction_folder, Channels=None, verbose=self.verbose)
                _ref_cls._load_image(sel_channels=[DriftChannel], ImSize=self.image_size,
                                     NbufferFrame=self.num_buffer_frames, NemptyFrame=self.num_empty_frames)
                RefImage = _own_ref = _ref_cls._dev.pop(DriftChannel)
            elif isinstance(RefImage, np.ndarray) and (np.array(RefImage.shape) == np.array(_DriftImage.shape)).all():
                if save_ref_im:
                    setattr(self, 'ref_im', RefImage)
            elif isinstance(RefImage, L.DeviceStack) and tuple(RefImage.shape) == tuple(_DriftImage.shape):
                pass
            else:
                raise ValueError("Wrong input of RefImage, should be either a matched sized image, or a filename")
            if precise_align:
                _drift, _drift_flag = align_image(_DriftImage, RefImage, use_autocorr=use_autocorr,
                                                  drift_channel=DriftChannel, verbose=self.verbose, **drift_kwargs)
            else:
                _drift, _error, _phasediff = phase_cross_correlation(RefImage, _DriftImage)
                _drift_flag = 2
        finally:
            if _own is not None:
                _own.free()
            if _own_ref is not None:
                _own_ref.free()
        if save_attr:
            setattr(self, 'drift_channel', DriftChannel)
            setattr(self, 'drift', _drift)
            setattr(self, 'drift_flag', _drift_flag)
        return _drift, _drift_flag

    def _warp_image(self, drift=None, correction_channels=None, corr_chromatic=True, chromatic_pf=None,
                    correction_folder=None, ref_channel=default_ref_channel, save_attrs=True, overwrite=False):
        """:850-965 — cubic ``map_coordinates`` at ``(grid - drift) + chromatic field``, mode 'nearest'."""
        import ctypes as C
        import warnings
        from imageanalysis3_amd import _lib as L
        from imageanalysis3_amd.io_tools.load import _as_buffer
        if drift is not None:
            _drift = np.array(drift)
        elif hasattr(self, 'drift'):
            _drift = getattr(self, 'drift')
        else:
            _drift = np.zeros(len(self.image_size))
            warnings.warn("drift not given to warp image. ")
        if correction_channels is None:
            correction_channels = self.loaded_channels
        _correction_channels = [str(_ch) for _ch in correction_channels]
        _chromatic_channels = [_ch for _ch in _correction_channels
                               if _ch != getattr(self, 'drift_channel', None) and _ch != getattr(self, 'dapi_channel', None)]
        _ch_2_finish_warp = {_ch: self.correction_log[_ch].get('corr_drift', False) or not _drift.any()
                             for _ch in _correction_channels}
        _ch_2_finish_chromatic = {_ch: self.correction_log[_ch].get('corr_chromatic', False) for _ch in _chromatic_channels}
        _logs = [_ch_2_finish_warp.get(_ch) and _ch_2_finish_chromatic.get(_ch, True) for _ch in _correction_channels]
        if np.array(_logs).all():
            if self.verbose:
                print("- Warp drift and chromatic already finished, skip. ")
            return
        if correction_folder is None:
            correction_folder = self.correction_folder
        if corr_chromatic and chromatic_pf is None:                          # :889-897
            from imageanalysis3_amd.io_tools.load import load_correction_profile
            chromatic_pf = load_correction_profile('chromatic', _chromatic_channels,
                                                   correction_folder=correction_folder, all_channels=self.channels,
                                                   ref_channel=ref_channel, im_size=self.image_size,
                                                   verbose=self.verbose)
        _done_chs, _stacks = [], []
        for _ch in _correction_channels:
            _finish_warp = _ch_2_finish_warp.get(_ch)
            _finish_chromatic = _ch_2_finish_chromatic.get(_ch, True)
            if _finish_warp and (_finish_chromatic or not corr_chromatic):
                continue
            if _ch not in self._dev:
                continue
            _d = np.zeros(3)
            if not _finish_warp:
                _d = np.ascontiguousarray(_drift, dtype=np.float64)
                self.correction_log[_ch]['corr_drift'] = True
            _field, _fdt = None, 16          # 16: coordinates formed as (grid - drift) + field
            if not _finish_chromatic and corr_chromatic:
                if not (chromatic_pf[_ch] is None and str(_ch) == ref_channel):
                    _fb = _as_buffer(chromatic_pf[_ch])
                    if _fb.arr.shape != (3,) + tuple(self._dev[_ch].shape):
                        raise IndexError(f"chromatic_pf[{_ch}] shape {_fb.arr.shape} should be {(3,) + tuple(self._dev[_ch].shape)}")
                    _field, _fdt = _fb.ptr, _fb.dtype_code + 16
                self.correction_log[_ch]['corr_chromatic'] = True
            _out = L.DeviceStack.empty(self._dev[_ch].shape, self._dev[_ch].dtype)
            L.check(L.lib().ia3_warp3d_dev(self._dev[_ch]._h, L.dptr(np.ascontiguousarray(_d, dtype=np.float64)), _field, _fdt,
                                           3, L.MODE_NEAREST, C.c_double(0.0), _out._h))
            _done_chs.append(_ch)
            _stacks.append(_out)
        return self._finish(_done_chs, _stacks, save_attrs, None)

    def _gaussian_highpass(self, correction_channels=None, gaussian_sigma=3, gaussian_truncate=2, save_attrs=True,
                           overwrite=False):
        """:967-1031."""
        import ctypes as C
        from imageanalysis3_amd import _lib as L
        if correction_channels is None:
            correction_channels = self.loaded_channels
        _correction_channels = [str(_ch) for _ch in correction_channels if str(_ch) != getattr(self, 'dapi_channel', None)]
        _logs = [self.correction_log[_ch].get('corr_highpass', False) for _ch in _correction_channels]
        if np.array(_logs).all():
            return
        _correction_channels = [_ch for _ch, _log in zip(_correction_channels, _logs) if not _log]
        w, r = L.gaussian_taps(gaussian_sigma, gaussian_truncate)
        _stacks = []
        for _ch in _correction_channels:
            if _ch not in self._dev:
                _stacks.append(None)
                continue
            _out = L.DeviceStack.empty(self._dev[_ch].shape, self._dev[_ch].dtype)
            L.check(L.lib().ia3_gaussian_highpass_dev(self._dev[_ch]._h, C.c_double(gaussian_sigma),
                                                      C.c_double(gaussian_truncate), L.dptr(w), int(r), _out._h))
            _stacks.append(_out)      # np.clip to the uint16 range + astype are no-ops on a uint16 result
        return self._finish(_correction_channels, _stacks, save_attrs, 'corr_highpass')

    def _fit_spots(self, fit_channels=None, th_seed=1000, num_spots=None, fitting_kwargs={}, save_attrs=True,
                   overwrite=False):
        """:1033-1091 — ``fit_fov_image`` on the resident stack of every fit channel."""
        from imageanalysis3_amd.spot_tools.fitting import fit_fov_image
        if fit_channels is None:
            fit_channels = self.loaded_channels
        _fit_channels = [str(_ch) for _ch in fit_channels
                         if str(_ch) != getattr(self, 'drift_channel', None) and str(_ch) != getattr(self, 'dapi_channel', None)]
        _fit_logs = [hasattr(self, f'spots_{_ch}') and not overwrite for _ch in _fit_channels]
        if np.array(_fit_logs).all():
            return
        _fit_channels = [_ch for _ch, _log in zip(_fit_channels, _fit_logs) if not _log]
        if isinstance(th_seed, (int, float)):
            _ch_2_thSeed = {_ch: th_seed for _ch in _fit_channels}
        elif isinstance(th_seed, dict):
            _ch_2_thSeed = {str(_ch): _th for _ch, _th in th_seed.items()}
        _spots_list = []
        for _ch in _fit_channels:
            if _ch not in self._dev:
                continue
            _spots = fit_fov_image(self._dev[_ch], _ch, th_seed=_ch_2_thSeed.get(_ch, default_seed_th),
                                   max_num_seeds=num_spots, verbose=self.verbose, **fitting_kwargs)
            _cell_ids = np.ones(len(_spots), dtype=np.int32) - 1
            if save_attrs:
                setattr(self, f"spots_{_ch}", _spots)
                setattr(self, f"spots_cell_ids_{_ch}", _cell_ids)
            else:
                _spots_list.append(_spots)
        if save_attrs:
            return
        return _spots_list

    def _fit_spots_by_segmentation(self, channel, seg_label, th_seed=500, num_spots=None, fitting_kwargs={},
                                   segment_search_radius=3, save_attrs=True, verbose=False):
        """:1093-1153 — per label of ``seg_label``: bounding box (shifted by the drift), ``fit_fov_image`` on that crop
        of the resident stack, keep the spots whose neighbourhood votes for the label.  ``seg_label`` is a host array
        (segmentation itself is out of scope); the crops are cut on the device."""
        from .cell import segmentation_mask_2_bounding_box
        from imageanalysis3_amd.spot_tools.fitting import fit_fov_image
        from .partition_spots import Spots_Partition
        drift = getattr(self, 'drift', np.zeros(len(self.image_size)))
        if self.verbose:
            print(f"- Start fitting spots in each segmentation")
        stack = self._dev[str(channel)]
        labels = np.unique(seg_label)
        tables, owners = [], []
        for label in labels[labels > 0]:
            mask = seg_label == label
            # the reference calls segmentation_mask_2_bounding_box(mask, 3): the 3 lands in `cell_id` (:1117), the margin
            # stays at its default of one voxel
            box = segmentation_mask_2_bounding_box(mask, 3).translate_drift(drift=drift)
            crop = stack.crop(box.array)
            try:
                rows = fit_fov_image(crop, str(channel), th_seed=th_seed, max_num_seeds=num_spots, verbose=verbose,
                                     **fitting_kwargs)
            finally:
                crop.free()
            if len(rows) == 0:
                continue
            rows = Spots3D(rows)
            rows[:, rows.coordinate_indices] = rows[:, rows.coordinate_indices] + box.array[:, 0]   # back to FOV coordinates
            inside = Spots_Partition.spots_to_labels(mask, rows, search_radius=segment_search_radius, verbose=False) > 0
            if inside.any():
                tables.append(rows[inside])
                owners.append(np.full(int(inside.sum()), label, dtype=np.int32))
        if tables:
            _all_spots, _all_cell_ids = np.concatenate(tables), np.concatenate(owners)
        else:
            _all_spots, _all_cell_ids = np.array([]), np.array([])
            print(f"No spots detected.")
        if save_attrs:
            setattr(self, f"spots_{channel}", _all_spots)
            setattr(self, f"spots_cell_ids_{channel}", _all_cell_ids)
            return
        return _all_spots, _all_cell_ids

    # saving / loading: empty in the reference as well (:1155-1164)
    def _save_to_hdf5(self):
        pass

    def _save_to_npy(self, save_channels, save_folder=None, save_basenames=None):
        if save_folder is None:
            pass

    def _load_from_hdf5(self):
        pass

    # -- file helpers -------------------------------------------------------------------------------------------------
    @staticmethod
    def _FindDaxChannels(dax_filename, verbose=True):
        """:1166-1182 — channel names from the shutter file named in the .xml."""
        import os
        import re
        import xml.etree.ElementTree as ET
        try:
            _hal_info = ET.parse(dax_filename.replace('.dax', '.xml')).getroot()
            _shutter_filename = _hal_info.findall('illumination/shutters')[0].text
            _names = os.path.basename(_shutter_filename).split(os.extsep)[0].split('_')
            return [_ch for _ch in _names if len(re.findall(r'^[0-9]+$', _ch))]
        except Exception:
            return None

    @staticmethod
    def _FindGlobalPosition(dax_filename, verbose=True):
        """:1184-1195 — stage position (micron) from the .xml next to the movie."""
        import xml.etree.ElementTree as ET
        try:
            _hal_info = ET.parse(dax_filename.replace('.dax', '.xml')).getroot()
            return np.array(_hal_info.findall('acquisition/stage_position')[0].text.split(','), dtype=np.float64)
        except Exception:
            raise ValueError(f"Positions not properly parsed")

    @staticmethod
    def _LoadSegmentation(segmentation_filename, fov_id=None, verbose=True):
        """:1234-1255 — label image from .npy / .pkl / .hdf5 (``<fov_id>/dna_mask``)."""
        import os
        import pickle
        if not isinstance(segmentation_filename, str) or not os.path.isfile(segmentation_filename):
            raise ValueError(f"invalid segmentation_filename: {segmentation_filename}")
        if verbose:
            print(f"-- load segmentation from: {segmentation_filename}")
        _ext = segmentation_filename.split(os.extsep)[-1]
        if _ext == 'npy':
            _seg_label = np.load(segmentation_filename)
        elif _ext == 'pkl':
            _seg_label = pickle.load(open(segmentation_filename, 'rb'))
        elif _ext == 'hdf5' or _ext == 'h5':
            from imageanalysis3_amd.io_tools import h5lite as h5py
            with h5py.File(segmentation_filename, 'r') as _f:
                if fov_id is None:
                    fov_id = list(_f.keys())[0]
                _seg_label = _f[str(fov_id)]['dna_mask'][:]
        return _seg_label

    @staticmethod
    def _LoadInfFile(inf_filename):
        """:1197-1205."""
        _info_dict = {}
        with open(inf_filename, 'r') as _info_hd:
            for _line in _info_hd.readlines():
                _key, _value = _line.rstrip().split(' = ')
                _info_dict[_key] = _value
        return _info_dict

    @staticmethod
    def _FindImageSize(dax_filename, channels=None, NbufferFrame=default_num_buffer_frames, verbose=True):
        """:1207-1232."""
        if channels is None:
            channels = DaxProcesser._FindDaxChannels(dax_filename)
        try:
            _info_dict = DaxProcesser._LoadInfFile(dax_filename.replace('.dax', '.inf'))
            _dx, _dy = (int(_v) for _v in _info_dict['frame dimensions'].split('x'))
            _dz = (int(_info_dict['number of frames']) - 2 * NbufferFrame) / len(channels)
            if _dz != int(_dz):
                raise ValueError("Wrong num_color, should be integer!")
            return np.array([int(_dz), _dx, _dy], dtype=np.int32)
        except Exception:
            return np.array(default_im_size)


def batch_process_image_quick(dax_filename, correction_folder, sel_channels, drift_channel='488', dapi_channel='405',
                              corr_hot_pixels=True, corr_illumination=True, verbose=True):
    """classes/preprocess.py:1257-1278 — load the selected channels of one movie, remove hot pixels, divide by the
    illumination profiles found in ``correction_folder``; returns the images."""
    _cls = DaxProcesser(dax_filename, correction_folder, Channels=None, DriftChannel=drift_channel,
                        DapiChannel=dapi_channel, verbose=verbose)
    _cls._load_image(sel_channels=sel_channels)
    if corr_hot_pixels:
        _cls._corr_hot_pixels_3D(correction_channels=sel_channels)
    if corr_illumination:
        _cls._corr_illumination(correction_channels=sel_channels)
    return [getattr(_cls, f"im_{_ch}") for _ch in sel_channels]
