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


def test_cubic_constant_mode_restatement_vs_scipy():
    """Order-3 ``map_coordinates`` with ``mode='constant'`` (warp_3d_image's default border mode with warp_order=3): the
    rules csrc/warp.hip implements — mirror-boundary prefilter without padding, cval outside [0, n-1], mirrored taps —
    restated in oracle/np_oracle.py and compared with SciPy bit for bit, lines of 2 ... 700 samples (z^(n-1) underflows
    beyond ~566)."""
    import scipy.ndimage as ndi
    import np_oracle as O
    rng = np.random.RandomState(3)
    for n in (2, 3, 4, 5, 17, 64, 566, 567, 700):
        x = rng.randint(0, 60000, size=n).astype(np.float64)
        assert np.array_equal(O.spline3_mirror_line(x), ndi.spline_filter1d(x, 3, mode="constant", output=np.float64)), n
    for n in (2, 3, 4, 7, 30):
        x = rng.randint(100, 5000, size=n).astype(np.float64)
        cs = np.concatenate([rng.uniform(-3, n + 2, size=300), np.arange(-2, n + 2).astype(float),
                             [n - 1 - 1e-12, 1e-12, -1e-12, n - 1 + 1e-9, 0.5, n - 1.5]])
        ref = ndi.map_coordinates(x, [cs], order=3, mode="constant", cval=-7.0, output=np.float64)
        assert np.array_equal(O.cubic_constant_1d(x, cs, -7.0), ref), n
