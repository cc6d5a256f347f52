"""Drop-in for the hot-path part of the reference's ``spot_tools/matching.py`` (:148-287): unique bead
pairing and Delaunay-neighbour outlier rejection.  Host logic on at most a few hundred fitted centres
per drift crop (the centres themselves come from the device fit)."""
import numpy as np


def find_paired_centers(tar_cts, ref_cts, drift=None,
                        cutoff=2, dimension=3,
                        return_paired_cts=True,
                        return_kept_inds=False,
                        verbose=False):
    """spot_tools/matching.py:148-222 — pairs (t, r) with |t - (r + drift)| <= cutoff that are unique in
    both directions; returns the mean shift of the pairs (+ the paired centres / indices)."""
    from scipy.spatial.distance import cdist
    _dimension = int(dimension)
    _tar_cts = np.array(tar_cts)
    _ref_cts = np.array(ref_cts)
    if np.shape(_tar_cts)[1] > 3:
        _tar_cts = _tar_cts[:, 1:1 + _dimension]
    if np.shape(_ref_cts)[1] > 3:
        _ref_cts = _ref_cts[:, 1:1 + _dimension]
    if drift is None:
        _drift = np.zeros(np.shape(_tar_cts)[1])
    else:
        _drift = np.array(drift, dtype=float)[:_dimension]
    if verbose:
        print(f"-- aligning {len(_tar_cts)} centers to {len(_ref_cts)} ref_centers, given drift:{np.round(_drift,2)}",
              end=', ')
    _close = cdist(_tar_cts, _ref_cts + _drift) <= cutoff
    _tar_inds, _ref_inds = np.where(_close)
    _tar_unique = np.sum(_close, axis=1) == 1
    _ref_unique = np.sum(_close, axis=0) == 1
    _sel = _tar_unique[_tar_inds] & _ref_unique[_ref_inds]
    _pairs = np.stack([_tar_inds[_sel], _ref_inds[_sel]], axis=1) if _sel.any() else np.zeros((0, 2), dtype=int)
    _paired_tar_cts = _tar_cts[_pairs[:, 0]] if len(_pairs) else np.array([])
    _paired_ref_cts = _ref_cts[_pairs[:, 1]] if len(_pairs) else np.array([])
    _new_drift = np.nanmean(_paired_tar_cts - _paired_ref_cts, axis=0)
    if verbose:
        print(f"{len(_paired_tar_cts)} pairs found, updated_drift:{np.round(_new_drift,2)}")
    _return_args = [_new_drift]
    if return_paired_cts:
        _return_args += [_paired_tar_cts, _paired_ref_cts]
    if return_kept_inds:
        _return_args += [np.array(_pairs[:, 0], dtype=int), np.array(_pairs[:, 1], dtype=int)]
    return tuple(_return_args)


def check_paired_centers(paired_tar_cts, paired_ref_cts,
                         outlier_sigma=1.5,
                         return_paired_cts=True,
                         verbose=False):
    """spot_tools/matching.py:224-287 — a pair is kept if its shift is within mean + sigma*std of the
    inverse-distance-weighted shift of its Delaunay neighbours."""
    from scipy.spatial import Delaunay
    _tar_cts = np.array(paired_tar_cts, dtype=float)
    _ref_cts = np.array(paired_ref_cts, dtype=float)
    _shifts = _tar_cts - _ref_cts
    if verbose:
        print(f"-- check {len(_tar_cts)} pairs of centers", end=', ')
    _simplices = Delaunay(_ref_cts).simplices.copy()
    _new_shifts = []
    for _i, _rc in enumerate(_ref_cts):
        _nb_ids = np.unique(_simplices[(_simplices == _i).any(1)])
        _nb_ids = _nb_ids[(_nb_ids != _i) & (_nb_ids != -1)]
        _nb_weights = 1 / np.linalg.norm(_ref_cts[_nb_ids] - _rc, axis=1)
        _new_shifts.append(np.dot(_shifts[_nb_ids].T, _nb_weights) / np.sum(_nb_weights))
    _new_shifts = np.array(_new_shifts)
    _diffs = np.linalg.norm(_new_shifts - _shifts, axis=1)
    _keep_flags = np.array(_diffs < np.mean(_diffs) + np.std(_diffs) * outlier_sigma)
    _kept_tar_cts = _tar_cts[_keep_flags]
    _kept_ref_cts = _ref_cts[_keep_flags]
    _new_drift = np.nanmean(_kept_tar_cts - _kept_ref_cts, axis=0)
    if verbose:
        print(f"{len(_kept_tar_cts)} pairs kept. new drift:{np.round(_new_drift,2)}")
    _return_args = [_new_drift]
    if return_paired_cts:
        _return_args += [_kept_tar_cts, _kept_ref_cts]
    return tuple(_return_args)
