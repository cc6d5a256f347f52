"""CPU: host-side logic of the shims that carries arithmetic of its own."""
import numpy as np


def test_score_at_percentile_equals_scipy_bit_for_bit():
    """get_seeds(use_percentile=True) thresholds with scipy.stats.scoreatpercentile (spot_tools/fitting.py:76); the shim
    restates its arithmetic on a partition instead of a full sort."""
    from scipy.stats import scoreatpercentile
    from imageanalysis3_amd.spot_tools.fitting import _score_at_percentile
    rng = np.random.RandomState(7)
    for dtype in (np.float32, np.uint16, np.float64):
        for shape in ((5, 9, 11), (30, 64, 64), (1, 1, 7)):
            a = rng.gamma(2.0, 300.0, size=shape).astype(dtype)
            for per in (95, 2.5, 99.5, 0.25, 98, 1.0, 50, 0, 100, 37.3):
                got, ref = _score_at_percentile(a, per), scoreatpercentile(a, per)
                assert type(got) is type(ref) and got == ref, (dtype, shape, per, got, ref)
    assert np.isnan(_score_at_percentile(np.zeros((0, 3, 3), np.float32), 95))
