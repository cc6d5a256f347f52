"""Bead pairing for drift alignment (interface of the reference's ``spot_tools/matching.py:148-287``): unique
nearest pairs and Delaunay-neighbour outlier rejection.  Host logic on at most a few hundred fitted centres per drift
crop; the centres themselves come from the device fit."""
import numpy as np


def _coords(points, ndim):
    """(n, ndim) coordinates from centres or from 11-column spot rows (columns 1..ndim)."""
    pts = np.array(points)
    return pts[:, 1:1 + ndim] if pts.shape[1] > 3 else pts


def find_paired_centers(tar_cts, ref_cts, drift=None,
                        cutoff=2, dimension=3,
                        return_paired_cts=True,
                        return_kept_inds=False,
                        verbose=False):
    """matching.py:148-222 — pairs (target t, reference r) with ``|t - (r + drift)| <= cutoff`` in which neither
    point has a second candidate; returns the mean shift of the pairs (then the paired centres, then the indices,
    as requested)."""
    from scipy.spatial.distance import cdist
    ndim = int(dimension)
    tar, ref = _coords(tar_cts, ndim), _coords(ref_cts, ndim)
    shift0 = np.zeros(tar.shape[1]) if drift is None else np.array(drift, dtype=float)[:ndim]
    if verbose:
        print(f"-- aligning {len(tar)} centers to {len(ref)} ref_centers, given drift:{np.round(shift0,2)}", end=', ')
    near = cdist(tar, ref + shift0) <= cutoff
    single_t = near.sum(axis=1) == 1          # targets with exactly one reference in range
    single_r = near.sum(axis=0) == 1          # references with exactly one target in range
    it, ir = np.nonzero(near & single_t[:, None] & single_r[None, :])
    if len(it):
        pt, pr = tar[it], ref[ir]
    else:
        pt, pr = np.array([]), np.array([])
    new_drift = np.nanmean(pt - pr, axis=0)
    if verbose:
        print(f"{len(pt)} pairs found, updated_drift:{np.round(new_drift,2)}")
    out = [new_drift]
    if return_paired_cts:
        out += [pt, pr]
    if return_kept_inds:
        out += [it.astype(int), ir.astype(int)]
    return tuple(out)


def check_paired_centers(paired_tar_cts, paired_ref_cts,
                         outlier_sigma=1.5,
                         return_paired_cts=True,
                         verbose=False):
    """matching.py:224-287 — every pair's shift is compared with the inverse-distance-weighted shift of its Delaunay
    neighbours (triangulation of the reference points); pairs whose deviation exceeds mean + ``outlier_sigma``·std are
    dropped.  Returns the mean shift of the kept pairs (and the kept centres)."""
    from scipy.spatial import Delaunay
    tar = np.array(paired_tar_cts, dtype=float)
    ref = np.array(paired_ref_cts, dtype=float)
    shifts = tar - ref
    if verbose:
        print(f"-- check {len(tar)} pairs of centers", end=', ')
    tets = Delaunay(ref).simplices
    expected = np.empty_like(shifts)
    for i in range(len(ref)):
        nb = np.unique(tets[(tets == i).any(axis=1)])
        nb = nb[(nb != i) & (nb != -1)]
        w = 1.0 / np.linalg.norm(ref[nb] - ref[i], axis=1)
        expected[i] = np.dot(shifts[nb].T, w) / np.sum(w)
    dev = np.linalg.norm(expected - shifts, axis=1)
    keep = np.array(dev < np.mean(dev) + np.std(dev) * outlier_sigma)
    new_drift = np.nanmean(tar[keep] - ref[keep], axis=0)
    if verbose:
        print(f"{int(keep.sum())} pairs kept. new drift:{np.round(new_drift,2)}")
    return (new_drift, tar[keep], ref[keep]) if return_paired_cts else (new_drift,)
