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
    """classes/__init__.py:57-88 — legacy per-cell fit.  For every chromosome coordinate: seeds within a radius
    (``visual_tools.get_seed_in_distance(_im, coord, *_seeding_args)``), ``Fitting_v3`` first fit (+ refit sweeps when
    ``_check_fitting``), heights divided by ``nanmedian(_im)`` when ``_normalization``.  One (N,11) array (or an
    empty array) per coordinate."""
    if _verbose:
        print(f"+++ fitting for region:{_id}")
    if _im is None:
        return [np.array([]) for _ in _chrom_coords]
    scale = np.nanmedian(_im) if _normalization else None
    tables = []
    for coord in _chrom_coords:
        seeds = visual_tools.get_seed_in_distance(_im, coord, *_seeding_args)
        if len(seeds) == 0:
            tables.append(np.array([]))
            continue
        fitter = Fitting_v3.iter_fit_seed_points(_im, seeds.T, *_fitting_args)
        fitter.firstfit()
        if _check_fitting:
            fitter.repeatfit()
        rows = np.array(fitter.ps)
        if scale is not None:
            rows[:, 0] /= scale
        tables.append(rows)
    return tables
