"""reference: io_tools/load.py — only ``find_image_background`` (:642-687) so far.

Host-side NumPy (a few thousand voxels per call).  SURVEY.md §8(f4) lists the batched
histogram-mode kernel for ``normalize_local`` as a "next" row; until it exists this helper is
the reference's own arithmetic on the host, outside the measured hot path.
"""
import numpy as np
from .. import _image_dtype


def find_image_background(im, dtype=_image_dtype, bin_size=10, make_plot=False, max_iter=10):
    """Histogram-peak background level (io_tools/load.py:642-687)."""
    import scipy.signal
    if dtype is None:
        dtype = im.dtype
    _cts, _bins = np.histogram(im, bins=np.arange(np.iinfo(dtype).min, np.iinfo(dtype).max, bin_size))
    _peaks = []
    _height = np.size(im) / 50
    _iter = 0
    while len(_peaks) == 0:
        _height = _height / 2
        _peaks, _params = scipy.signal.find_peaks(_cts, height=_height)
        _iter += 1
        if _iter > max_iter:
            break
    if _iter > max_iter:
        _background = np.nanmedian(im)
    else:
        _sel_peak = _peaks[np.argmax(_params['peak_heights'])]
        _background = (_bins[_sel_peak] + _bins[_sel_peak + 1]) / 2
    return _background
