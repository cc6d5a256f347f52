"""Drop-in for the hot-path part of the reference's ``correction_tools/alignment.py``:
``generate_drift_crops`` (:87-135), ``align_beads`` (:139-216), ``align_image`` (:527-695), plus
``phase_cross_correlation`` (scikit-image's function as the reference calls it at :631) on the device.

Both stacks are uploaded once; the <= 8 drift crops are cut on the device (``ia3_stack_crop``) and each
crop is aligned by rocFFT phase correlation + upsampled DFT, or by bead fitting + FFT rough shift.
"""
import ctypes as C
import time
import numpy as np

from .. import _lib as L
from .. import _allowed_colors, _image_size, _num_buffer_frames, _num_empty_frames, _correction_folder

# The reference calls skimage.registration.phase_cross_correlation without pinning a version.  None = the
# un-normalised correlation of scikit-image 0.17 / 0.18 — the release found next to the reference's other dependencies
# in this image (0.18.3), against which this path is pinned bit for bit (tests/golden/phase.npz); "phase" = the
# default of scikit-image >= 0.19, checked against the oracle's restatement of the published algorithm only.
DEFAULT_NORMALIZATION = None


def _find_boundary(_ct, _radius, _im_size):
    """correction_tools/alignment.py:80-85."""
    return np.array([[max(_c - _radius, 0), min(_c + _radius, _sz)] for _c, _sz in zip(_ct, _im_size)], dtype=int)


def generate_drift_crops(single_im_size=_image_size, coord_sel=None, drift_size=None):
    """correction_tools/alignment.py:87-135 — 8 crops (8,3,2) around quarter/half anchor points."""
    _single_im_size = np.array(single_im_size)
    if coord_sel is None:
        coord_sel = np.array(_single_im_size / 2, dtype=int)
    if coord_sel[-2] >= _single_im_size[-2] or coord_sel[-1] >= _single_im_size[-1]:
        raise ValueError(f"wrong input coord_sel:{coord_sel}, should be smaller than single_im_size:{single_im_size}")
    if drift_size is None:
        drift_size = int(np.max(_single_im_size) / 4)
    _cz, _cx, _cy = coord_sel[-3] / 2, coord_sel[-2], coord_sel[-1]
    _sx, _sy = _single_im_size[-2], _single_im_size[-1]
    crop_cts = [
        (_cz, _cx / 2, _cy / 2), (_cz, (_cx + _sx) / 2, (_cy + _sy) / 2),
        (_cz, (_cx + _sx) / 2, _cy / 2), (_cz, _cx / 2, (_cy + _sy) / 2),
        (_cz, _cx, _cy / 2), (_cz, _cx, (_cy + _sy) / 2),
        (_cz, _cx / 2, _cy), (_cz, (_cx + _sx) / 2, _cy),
    ]
    return np.array([_find_boundary(_ct, _radius=drift_size / 2, _im_size=single_im_size) for _ct in crop_cts])


def phase_cross_correlation(reference_image, moving_image, upsample_factor=1, normalization="default", **kwargs):
    """skimage.registration.phase_cross_correlation for 3-D stacks on the device (rocFFT).
    Returns (shift, error, phasediff).  Inputs: ndarrays (uint16/float32) or DeviceStacks."""
    if normalization == "default":
        normalization = DEFAULT_NORMALIZATION
    if normalization not in ("phase", None):
        raise ValueError("normalization must be either 'phase' or None")
    _norm = 1 if normalization == "phase" else 0
    shift = (C.c_double * 3)()
    err, ph = C.c_double(0), C.c_double(0)
    if isinstance(reference_image, L.DeviceStack) and isinstance(moving_image, L.DeviceStack):
        L.check(L.lib().ia3_phase_xcorr3d_dev(reference_image._h, moving_image._h, int(upsample_factor), _norm,
                                              shift, C.byref(err), C.byref(ph)))
    else:
        a, b = L.as_stack_array(reference_image), L.as_stack_array(moving_image)
        if a.shape != b.shape:
            raise ValueError("images must be same shape")
        if a.dtype != b.dtype:
            b = b.astype(a.dtype)
        L.check(L.lib().ia3_phase_xcorr3d(L.ptr(a), L.ptr(b), L.dtype_code(a), a.shape[0], a.shape[1], a.shape[2],
                                          int(upsample_factor), _norm, shift, C.byref(err), C.byref(ph)))
    return np.array([shift[0], shift[1], shift[2]]), float(err.value), float(ph.value)


class DriftReference(object):
    """The reference bead image of a run with the half spectra of its drift crops kept on the device
    (``ia3_drift_ref``): every image of a run is aligned to the same reference (classes/batch_functions.py:169-206), so
    its transforms are made once.  Hand it to ``align_image`` as ``ref_im``; drifts are those of the plain image.
    ``ref_im``: (Z,X,Y) ndarray or resident ``DeviceStack``; ``crop_list``: as for ``align_image`` (default:
    ``generate_drift_crops`` of the image size)."""

    def __init__(self, ref_im, crop_list=None, dtype=np.uint16):
        self._own = not isinstance(ref_im, L.DeviceStack)
        if self._own and not isinstance(ref_im, np.ndarray):
            raise IOError(f"Wrong input file type, {type(ref_im)} should be np.ndarray or a resident stack")
        self.stack = L.DeviceStack.upload(ref_im if ref_im.dtype == np.dtype(dtype) else ref_im.astype(dtype)) if self._own else ref_im
        self.shape, self.dtype = tuple(self.stack.shape), self.stack.dtype
        if crop_list is None:
            crop_list = generate_drift_crops(self.shape)
        lims = np.array(crop_list, dtype=int).reshape(-1, 3, 2).copy()
        lims[:, :, 0] = np.maximum(lims[:, :, 0], 0)
        lims[:, :, 1] = np.minimum(lims[:, :, 1], np.array(self.shape)[None, :])
        self.crops = np.ascontiguousarray(lims, dtype=np.int32)
        h = C.c_void_p()
        L.check(L.lib().ia3_drift_ref_create(self.stack._h, self.crops.ctypes.data_as(C.POINTER(C.c_int)), len(self.crops),
                                             C.byref(h)))
        self._h = h

    def free(self):
        if getattr(self, "_h", None) is not None:
            L.lib().ia3_drift_ref_free(self._h)
            self._h = None
            if self._own:
                self.stack.free()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def align_beads(tar_cts, ref_cts,
                tar_im=None, ref_im=None,
                use_fft=True, fft_filt_size=0,
                match_distance_th=2.,
                check_paired_cts=True,
                outlier_sigma=1.5,
                return_paired_cts=True,
                verbose=True):
    """correction_tools/alignment.py:139-216 — mean shift of uniquely paired bead centres after an FFT
    rough alignment of the two crops.  Returns (drift,) or (drift, paired target centres, paired reference centres)."""
    from ..alignment_tools import fft3d_from2d, translation_align_pts
    from ..spot_tools.matching import find_paired_centers, check_paired_centers
    if not use_fft:
        # no images: translation from the point sets alone (:177-185; the reference names the returned point sets
        # reference first, target second)
        drift, pair_ref, pair_tar = translation_align_pts(np.array(ref_cts), np.array(tar_cts),
                                                          cutoff=float(match_distance_th), return_pts=True,
                                                          verbose=verbose)
    else:
        if tar_im is None or ref_im is None:
            raise ValueError("both tar_im and ref_im should be given if use FFT!")
        if np.shape(tar_im) != np.shape(ref_im):
            raise IndexError(f"tar_im shape:{np.shape(tar_im)} should match ref_im shape:{np.shape(ref_im)}")
        # integer shift from the projections' cross-correlation, searched over half the crop
        coarse = fft3d_from2d(tar_im, ref_im, gb=fft_filt_size, max_disp=np.max(np.shape(tar_im)) / 2)
        # beads that pair up uniquely within the matching distance once the coarse shift is applied
        drift, pair_tar, pair_ref = find_paired_centers(np.array(tar_cts), np.array(ref_cts), coarse,
                                                        cutoff=float(match_distance_th), return_paired_cts=True,
                                                        verbose=verbose)
    if verbose:
        print(f"-- {len(pair_ref)} bead pairs, drift before the outlier test: {drift}")
    if check_paired_cts and len(pair_ref) > 3:   # neighbour-consistency test needs a triangulation: at least 4 pairs
        drift, pair_tar, pair_ref = check_paired_centers(pair_tar, pair_ref, outlier_sigma=outlier_sigma,
                                                         return_paired_cts=True, verbose=verbose)
    return (drift, pair_tar, pair_ref) if return_paired_cts else (drift,)


def _consensus_drift(drifts, min_good_drifts, drift_diff_th):
    """alignment.py:664-674: once `min_good_drifts` crops are in, the crops within `drift_diff_th` of the running mean;
    if enough of them agree their mean is the answer, else None."""
    d = np.asarray(drifts, dtype=np.float64)
    if len(d) < min_good_drifts:
        return None, None
    centre = np.nanmean(d, axis=0)
    close = np.flatnonzero(np.linalg.norm(d - centre, axis=1) <= drift_diff_th)
    if len(close) < min_good_drifts:
        return None, None
    return np.nanmean(d[close], axis=0), close


def _closest_three_drift(drifts):
    """alignment.py:676-693: no consensus — the two crops that agree best and the crop nearest to both of them."""
    from scipy.spatial.distance import pdist, squareform
    d = np.asarray(drifts, dtype=np.float64)
    gaps = squareform(pdist(d))
    np.fill_diagonal(gaps, np.inf)
    pair = np.array(np.unravel_index(np.argmin(gaps), gaps.shape))
    third = int(np.argmin(gaps[:, pair].sum(1)))
    return np.nanmean(np.concatenate([d[pair], d[third:third + 1]]), axis=0)


_default_align_corr_args = {
    'single_im_size': _image_size, 'num_buffer_frames': _num_buffer_frames, 'num_empty_frames': _num_empty_frames,
    'correction_folder': _correction_folder, 'illumination_corr': True, 'bleed_corr': False,
    'chromatic_corr': False, 'z_shift_corr': False, 'hot_pixel_corr': True, 'normalization': False,
}

_default_align_fitting_args = {
    'th_seed': 300, 'th_seed_per': 95, 'use_percentile': False, 'use_dynamic_th': True,
    'min_dynamic_seeds': 10, 'max_num_seeds': 200,
}


def align_image(
        src_im: np.ndarray,
        ref_im: np.ndarray,
        crop_list=None,
        use_autocorr=True, precision_fold=100,
        min_good_drifts=3, drift_diff_th=1.,
        all_channels=_allowed_colors,
        ref_all_channels=None,
        drift_channel='488',
        correction_args={},
        fitting_args={},
        match_distance_th=2.,
        verbose=True,
        detailed_verbose=False,
):
    """correction_tools/alignment.py:527-695 — per-crop sub-pixel drift + 3-of-n consensus.
    Returns (drift (3,) float64, flag): drift = ref - src (add to source coordinates); flag 1 = the
    crops disagreed and the mean of the three mutually closest drifts is returned.

    ``src_im`` / ``ref_im``: ndarrays, or .dax filenames which are corrected through
    ``io_tools.load.correct_fov_image`` with ``correction_args`` (pass the illumination profile there)."""
    from ..spot_tools.fitting import fit_fov_image, select_sparse_centers
    _correction_args = {_k: _v for _k, _v in _default_align_corr_args.items()}
    _correction_args.update(correction_args)
    _fitting_args = {_k: _v for _k, _v in _default_align_fitting_args.items()}
    _fitting_args.update(fitting_args)
    _all_channels0 = [str(_ch) for _ch in all_channels]
    _ref_all_channels0 = _all_channels0 if ref_all_channels is None else [str(_ch) for _ch in ref_all_channels]

    def _load(_f, _chs):                                                     # :577-607
        import os
        from ..io_tools.load import correct_fov_image
        if not os.path.isfile(_f) or _f.split('.')[-1] != 'dax':
            raise IOError(f"input image: {_f} should be a .dax file!")
        return correct_fov_image(_f, [str(drift_channel)], all_channels=_chs, calculate_drift=False,
                                 return_drift=False, verbose=detailed_verbose,
                                 **{_k: _v for _k, _v in _correction_args.items() if _k != 'correction_folder'})[0][0]
    if isinstance(src_im, str):
        src_im = _load(src_im, _all_channels0)
    if isinstance(ref_im, str):
        ref_im = _load(ref_im, _ref_all_channels0)
    _dref = ref_im if isinstance(ref_im, DriftReference) else None   # reference crop spectra kept on the device
    if _dref is not None:
        if not use_autocorr:
            raise ValueError("a DriftReference holds phase-correlation spectra: use_autocorr=True")
        if crop_list is not None and not np.array_equal(np.array(crop_list, dtype=int).reshape(-1, 3, 2), _dref.crops):
            raise ValueError("crop_list differs from the crops the DriftReference was made for")
        crop_list = _dref.crops
        ref_im = _dref.stack
    _ok = (np.ndarray, L.DeviceStack)   # a DeviceStack (e.g. from correct_fov_image(return_device=True)) skips the upload
    if not isinstance(src_im, _ok) or not isinstance(ref_im, _ok):
        raise IOError(f"Wrong input file type, {type(src_im)} / {type(ref_im)} should be .dax file or np.ndarray")
    if np.shape(src_im) != np.shape(ref_im):
        raise IndexError(f"shape of target image:{np.shape(src_im)} and reference image:{np.shape(ref_im)} doesnt match!")
    if crop_list is None:
        _size = correction_args.get('single_im_size', np.shape(src_im))
        crop_list = generate_drift_crops(_size)
    for _crop in crop_list:
        if np.shape(np.array(_crop)) != (3, 2):
            raise IndexError("crop should be 3x2 np.ndarray.")
    _all_channels = [str(_ch) for _ch in all_channels]
    if str(drift_channel) not in _all_channels:
        raise ValueError(f"bead channel {drift_channel} not exist in all channels given:{_all_channels}")
    if verbose:
        print("-- start aligning given source image to given reference image.")
    _own_src, _own_ref = not isinstance(src_im, L.DeviceStack), not isinstance(ref_im, L.DeviceStack)
    _src = L.DeviceStack.upload(src_im) if _own_src else src_im
    if _own_ref:
        _ref = L.DeviceStack.upload(ref_im if ref_im.dtype == _src.dtype else ref_im.astype(_src.dtype))
    else:
        _ref = ref_im
        if _ref.dtype != _src.dtype:
            raise TypeError("resident source and reference stacks must have the same dtype")
    _result_flag = 0
    _drifts = []
    _updated_mean_dft = None
    try:
        if use_autocorr:
            # the crop loop, the phase correlations and the consensus rule in one library call (csrc/movie.cpp,
            # ia3_align_image_dev: the same entry the movie pipeline uses); drifts of the crops it needed come back
            _start_time = time.time()
            _lims = np.array(crop_list, dtype=int).reshape(-1, 3, 2).copy()
            _lims[:, :, 0] = np.maximum(_lims[:, :, 0], 0)
            _lims[:, :, 1] = np.minimum(_lims[:, :, 1], np.array(_src.shape)[None, :])
            _cl = np.ascontiguousarray(_lims, dtype=np.int32)
            _out, _flag, _nused = (C.c_double * 3)(), C.c_int(0), C.c_int(0)
            _each = np.zeros((len(_cl), 3), dtype=np.float64)
            if _dref is not None:
                L.check(L.lib().ia3_align_image_ref(_src._h, _dref._h, int(precision_fold),
                                                    1 if DEFAULT_NORMALIZATION == "phase" else 0, int(min_good_drifts),
                                                    C.c_double(float(drift_diff_th)), _out, C.byref(_flag), L.dptr(_each),
                                                    C.byref(_nused)))
            else:
                L.check(L.lib().ia3_align_image_dev(_src._h, _ref._h, _cl.ctypes.data_as(C.POINTER(C.c_int)), len(_cl),
                                                    int(precision_fold), 1 if DEFAULT_NORMALIZATION == "phase" else 0,
                                                    int(min_good_drifts), C.c_double(float(drift_diff_th)), _out,
                                                    C.byref(_flag), L.dptr(_each), C.byref(_nused)))
            _drifts = [_each[_i].copy() for _i in range(_nused.value)]
            if verbose:
                for _i, _dft in enumerate(_drifts):
                    print(f"-- drift {_i}: {np.around(_dft, 2)}")
                print(f"-- {len(_drifts)} crops in {time.time()-_start_time:.3f}s.")
            if _flag.value == 0:
                _updated_mean_dft = np.array([_out[0], _out[1], _out[2]])
                if verbose:
                    print(f"--- crops agree within {drift_diff_th} px: done.")
        for _i, _crop in enumerate(crop_list if not use_autocorr else []):
            _start_time = time.time()
            _lims = np.array(_crop, dtype=int)
            _sim, _rim = _src.crop(_lims), _ref.crop(_lims)
            try:
                _src_spots = fit_fov_image(_sim, drift_channel, verbose=detailed_verbose, **_fitting_args)
                _sp_src_cts = select_sparse_centers(_src_spots[:, 1:4], match_distance_th)
                _ref_spots = fit_fov_image(_rim, drift_channel, verbose=detailed_verbose, **_fitting_args)
                _sp_ref_cts = select_sparse_centers(_ref_spots[:, 1:4], match_distance_th, verbose=detailed_verbose)
                _dft, _paired_src_cts, _paired_ref_cts = align_beads(
                    _sp_src_cts, _sp_ref_cts, _sim, _rim, use_fft=True, match_distance_th=match_distance_th,
                    return_paired_cts=True, verbose=detailed_verbose)
                _dft = _dft * -1  # beads center is the opposite as cross correlation (:658)
            finally:
                _sim.free()
                _rim.free()
            _drifts.append(_dft)
            if verbose:
                print(f"-- drift {_i}: {np.around(_dft, 2)} in {time.time()-_start_time:.3f}s.")
            _updated_mean_dft, _agree = _consensus_drift(_drifts, min_good_drifts, drift_diff_th)
            if _updated_mean_dft is not None:
                if verbose:
                    print(f"--- crops {_agree} agree within {drift_diff_th} px: done.")
                break
    finally:
        if _own_src:
            _src.free()
        if _own_ref:
            _ref.free()
    if _updated_mean_dft is None:
        if verbose:
            print("-- the crops do not agree: mean of the closest three, flagged")
        _updated_mean_dft = _closest_three_drift(_drifts)
        _result_flag += 1
    return _updated_mean_dft, _result_flag
