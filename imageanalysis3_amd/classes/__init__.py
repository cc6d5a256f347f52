"""classes — containers the hot path hands back (reference: classes/preprocess.py:13-251) and the legacy
per-cell fitter ``_fit_single_image`` (reference: classes/__init__.py:57-88)."""
import numpy as np

from .. import visual_tools
from ..External import Fitting_v3

# classes/__init__.py:22-34 — data types a FOV save file may hold, and the default spot-table length
_allowed_kwds = {'combo': 'c',
                 'decoded': 'd',
                 'unique': 'u',
                 'relabeled_combo': 'l',
                 'relabeled_unique': 'v',
                 'merfish': 'm',
                 'rna': 'r',
                 'gene': 'g',
                 'protein': 'p',
                 }
_max_num_seeds = 4000
_min_num_seeds = 50


def _fit_single_image(_im, _id, _chrom_coords, _seeding_args, _fitting_args, _check_fitting=True,
                      _normalization=True, _verbose=False):
    """classes/__init__.py:57-88 — for every chromosome coordinate: seeds within a radius
    (``visual_tools.get_seed_in_distance(_im, coord, *_seeding_args)``), Fitting_v3 first fit (+ refit sweeps when
    ``_check_fitting``), heights divided by ``nanmedian(_im)`` when ``_normalization``.
    Returns a list with one (N,11) array (or an empty array) per coordinate."""
    if _verbose:
        print(f"+++ fitting for region:{_id}")
    _spots_for_chrom = []
    if _normalization:
        _norm_cst = np.nanmedian(_im)
    for _chrom_coord in _chrom_coords:
        if _im is None:
            _spots_for_chrom.append(np.array([]))
        else:
            _seeds = visual_tools.get_seed_in_distance(_im, _chrom_coord, *_seeding_args)
            if len(_seeds) == 0:
                _spots_for_chrom.append(np.array([]))
                continue
            _fitter = Fitting_v3.iter_fit_seed_points(_im, _seeds.T, *_fitting_args)
            _fitter.firstfit()
            if _check_fitting:
                _fitter.repeatfit()
            _spots = np.array(_fitter.ps)
            if _normalization:
                _spots[:, 0] = _spots[:, 0] / _norm_cst
            _spots_for_chrom.append(_spots)
    return _spots_for_chrom
