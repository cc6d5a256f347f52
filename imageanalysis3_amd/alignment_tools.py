"""Drop-in for the hot-path part of the reference's ``alignment_tools.py`` (:278-353):
pixel-level drift from max-projections by FFT cross-correlation, on the device (fft_align.hip)."""
import ctypes as C
import numpy as np

from . import _lib as L


def blurnorm2d(im, gb):
    """alignment_tools.py:278-283 needs ``cv2.blur``; OpenCV is an optional dependency of the reference
    used only when ``gb > 1`` (off the default path, ``fft_filt_size=0`` at correction_tools/alignment.py:141)."""
    import cv2  # noqa: F401  (raises ImportError where the reference would)
    im_ = im.astype(np.float32)
    return im_ / cv2.blur(im_, (gb, gb))


def fftalign_2d(im1, im2, center=[0, 0], max_disp=150, plt_val=False):
    """alignment_tools.py:286-328 — (xt, yt) of the windowed peak of the normalised cross-correlation."""
    a = np.ascontiguousarray(im1, dtype=np.float64)
    b = np.ascontiguousarray(im2, dtype=np.float64)
    if a.ndim != 2 or b.ndim != 2:
        raise IndexError("fftalign_2d takes 2-D images")
    c = np.ascontiguousarray(center, dtype=np.float64)
    out = (C.c_int * 2)()
    L.check(L.lib().ia3_fftalign_2d(L.dptr(a), a.shape[0], a.shape[1], L.dptr(b), b.shape[0], b.shape[1],
                                    L.dptr(c), C.c_double(float(max_disp)), out))
    return int(out[0]), int(out[1])


def fft3d_from2d(im1, im2, gb=5, max_disp=150):
    """alignment_tools.py:330-353 — integer [tz, tx, ty]; max-projections and FFTs on the device.
    ``im1``/``im2`` may be ndarrays or DeviceStacks."""
    if gb > 1:
        # cv2-normalised variant: projections on the host exactly as the reference, FFT on the device
        i1, i2 = _host(im1), _host(im2)
        im1_ = blurnorm2d(np.max(i1, 0), gb)
        im2_ = blurnorm2d(np.max(i2, 0), gb)
        tx, ty = fftalign_2d(im1_, im2_, center=[0, 0], max_disp=max_disp)
        sx, sy = im1_.shape
        im1_t = blurnorm2d(np.max(i1[:, max(tx, 0):sx + tx, max(ty, 0):sy + ty], axis=-1), gb)
        im2_t = blurnorm2d(np.max(i2[:, max(-tx, 0):sx - tx, max(-ty, 0):sy - ty], axis=-1), gb)
        tz, _ = fftalign_2d(im1_t, im2_t, center=[0, 0], max_disp=max_disp)
        return np.array([tz, tx, ty])
    out = (C.c_int * 3)()
    if isinstance(im1, L.DeviceStack) and isinstance(im2, L.DeviceStack):
        L.check(L.lib().ia3_fft3d_from2d_dev(im1._h, im2._h, C.c_double(float(max_disp)), out))
    else:
        a, b = L.as_stack_array(_host(im1)), L.as_stack_array(_host(im2))
        if a.shape != b.shape or a.dtype != b.dtype:
            raise IndexError("fft3d_from2d needs two stacks of the same shape and dtype")
        L.check(L.lib().ia3_fft3d_from2d(L.ptr(a), L.ptr(b), L.dtype_code(a), a.shape[0], a.shape[1], a.shape[2],
                                         C.c_double(float(max_disp)), out))
    return np.array([out[0], out[1], out[2]])


def _host(im):
    return im.download() if isinstance(im, L.DeviceStack) else im


def translation_align_pts(cents_fix, cents_target, cutoff=2., xyz_res=1,
                          plt_val=False, return_pts=False, verbose=False):
    """alignment_tools.py:356-419 — translation between two point sets without any image: every pair of target
    points votes, through the pairs of fixed points that are equally far apart (within ``cutoff``), for the
    translations that would map the pair's ends onto the first point of such a fixed pair; the fullest cell of
    the 3-D histogram of votes (cells of ``xyz_res``) gives a rough translation, and the median offset of the
    points that then pair up within ``2 * xyz_res`` is returned (with the paired fixed / target points if
    ``return_pts``).  Host NumPy/SciPy like the reference (used by ``align_beads(use_fft=False)``)."""
    from scipy.spatial.distance import pdist, cdist
    fix = np.array(cents_fix)
    tar = np.array(cents_target)
    if plt_val:
        raise NotImplementedError("plt_val=True (matplotlib figures) is not provided")
    i_fix, j_fix = np.triu_indices(len(fix), 1)          # pairs in the order of pdist / itertools.combinations
    i_tar, j_tar = np.triu_indices(len(tar), 1)
    d_fix, d_tar = pdist(fix), pdist(tar)
    votes = []
    for dt, a, b in zip(d_tar, i_tar, j_tar):
        first = fix[i_fix[np.abs(d_fix - dt) < cutoff]]  # first point of every fixed pair of that length
        votes.append(first - tar[a])
        votes.append(first - tar[b])
    votes = np.concatenate(votes) if votes else np.zeros((0, fix.shape[1]))
    nbins = np.array((np.max(votes, axis=0) - np.min(votes, axis=0)) / float(xyz_res), dtype=int)
    hist, edges = np.histogramdd(votes, bins=nbins)
    best = np.unravel_index(np.argmax(hist), hist.shape)
    rough = np.array([e[k] for e, k in zip(edges, best)])
    nearest = np.argmin(cdist(fix, tar + rough), axis=1)   # for every fixed point its closest shifted target point
    ok = np.sqrt(np.sum((tar[nearest] + rough - fix) ** 2, axis=-1)) < 2 * xyz_res
    pair_fix, pair_tar = fix[ok], tar[nearest[ok]]
    if len(pair_fix) == 0:
        raise ValueError("No matched points exist in cents[inds_closestF]")
    shift = np.median(pair_tar - pair_fix, axis=0)
    if verbose:
        print(f"--- {len(pair_fix)} points are aligned")
    if return_pts:
        return shift, pair_fix, pair_tar
    return shift
