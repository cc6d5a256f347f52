"""Drop-in for the part of the reference's ``External/Fitting_v3.py`` the legacy per-cell path uses:
``iter_fit_seed_points`` (:312-425; called from ``classes/__init__.py:57-88 _fit_single_image``).

Same device kernels as Fitting_v4 (fit.hip), run with ``model_variant = 1``: per-axis start widths
(``init_w`` is a 3-vector, default the package global ``_sigma_zxy``; Fitting_v3.py:71-79 including its
range test against the un-squared bounds), the v3 ``to_center`` (:81-87 — its third coordinate is
``2*delta*exp(-c1_)/(1+exp(-c2_))``, reproduced as written) and MINPACK's default ``maxfev``.
Voronoi ties go to the lowest seed index, which is exactly what v3's ``closest`` (cdist + argmin,
:39-47) does.  ``weight_sigma`` must be 0 (the reference default): its L1 width penalty is not built.
"""
import ctypes as C
import numpy as np

from .. import _lib as L
from .. import _sigma_zxy
from . import Fitting_v4 as _v4


def in_dim(x, y, z, xmax, ymax, zmax):
    """External/Fitting_v3.py:308-310."""
    return _v4.in_dim(x, y, z, xmax, ymax, zmax)


class iter_fit_seed_points(_v4.iter_fit_seed_points):
    def __init__(self, im, centers, radius_fit=5, min_delta_center=1., max_delta_center=2.5, n_max_iter=10,
                 max_dist_th=0.1, init_w=_sigma_zxy, weight_sigma=0):
        if weight_sigma:
            raise NotImplementedError("weight_sigma != 0 (Fitting_v3.py:124-132) is not supported on the device path")
        _v4.iter_fit_seed_points.__init__(self, im, centers, radius_fit=radius_fit,
                                          min_delta_center=min_delta_center, max_delta_center=max_delta_center,
                                          n_max_iter=n_max_iter, max_dist_th=max_dist_th)
        self.init_w = init_w
        self.weight_sigma = weight_sigma

    def _fit_params(self):
        return L.make_fit_params(self.radius_fit, self.min_delta_center, self.max_delta_center, self.n_max_iter,
                                 self.max_dist_th, 0.5, 4., 1.5, model_variant=1,
                                 init_w_zxy=np.asarray(self.init_w, dtype=np.float64)[:3])

    def firstfit(self):
        """External/Fitting_v3.py:340-383."""
        if len(self.centers) > 0:
            _v4.iter_fit_seed_points.firstfit(self)
        else:
            raise ValueError(f"{len(self.centers)} points have been seeded, exit.")
