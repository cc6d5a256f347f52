"""The certificate behind the one-pass spline prefilter of the cubic warp (csrc/warp.hip, spline_iir_strided_1p_k /
spline_iir_contig_k; reference: scipy.ndimage.map_coordinates as called at correction_tools/translate.py:27-30), checked
in IEEE float64 on the host with the same operations the kernels use.

Claim: the rounded anticausal recursion v[i] = fl(z * fl(v[i+1] - c[i])), z = sqrt(3) - 2 < 0, is monotone in v[i+1].
Two chains started K samples behind a tile at +B and -B with |v| <= B therefore enclose the true value at every
index, and where both hold the same bit pattern the true value has it too.  The kernels rely on exactly this and fall
back to the two-sweep recursion where the chains do not meet."""
import numpy as np
import pytest

Z = -0.26794919243112270647
GAIN = (1.0 - Z) * (1.0 - 1.0 / Z)


def causal(x, first):
    c = np.empty(len(x))
    c[0] = first
    for i in range(1, len(x)):
        c[i] = x[i] * GAIN + Z * c[i - 1]
    return c


def anticausal_exact(c):
    v = np.empty(len(c))
    v[-1] = c[-1] * (Z / (Z - 1.0))
    for i in range(len(c) - 2, -1, -1):
        v[i] = Z * (v[i + 1] - c[i])
    return v


def chains(c, start, stop, bound):
    """Both bounding chains from index start down to index stop (inclusive); they stand for v[start + 1] = +-bound."""
    pu, pl = bound, -bound
    for i in range(start, stop - 1, -1):
        pu = Z * (pu - c[i])
        pl = Z * (pl - c[i])
    return pu, pl


def _lines():
    rng = np.random.RandomState(7)
    yield "noise_u16", rng.randint(100, 5000, 900).astype(np.float64), 65535.0
    yield "dim_u16", rng.randint(90, 130, 900).astype(np.float64), 65535.0
    yield "beads", np.where(rng.rand(900) < 0.01, 60000.0, rng.randint(100, 400, 900)), 65535.0
    a = rng.uniform(0, 1, 900).astype(np.float32).astype(np.float64)
    yield "unit_f32", a, float(a.max())
    b = rng.uniform(-3e4, 3e4, 900).astype(np.float32).astype(np.float64)
    yield "signed", b, float(np.abs(b).max())
    d = rng.randint(100, 5000, 900).astype(np.float64)
    d[300:420] = 0.0
    yield "zero_stretch", d, 65535.0
    e = np.full(900, 1234.0)
    yield "constant", e, 65535.0
    f = rng.uniform(1e-30, 1e-28, 900)
    f[500:] = rng.uniform(1e20, 1e24, 400)
    yield "tiny_head_huge_tail", f, 1e24


@pytest.mark.parametrize("name,x,amax", list(_lines()), ids=[n for n, _, _ in _lines()])
def test_enclosure_and_agreement(name, x, amax):
    n = len(x)
    c = causal(x, x[0] * GAIN * 1.2)          # any start value: the claim is about the anticausal pass
    v = anticausal_exact(c)
    bound = 1.001 * GAIN * 3.0 * amax         # IirInit::bound of pass 1
    assert np.all(np.abs(v) <= bound)
    met = failed = 0
    for K in (8, 16, 32, 48, 52, 64):
        for a in range(0, n - K - 13, 12):    # tiles [a, a + 12), warm-up [a + 12, a + 12 + K)
            start, stop = a + 12 + K - 1, a + 12
            if start + 1 >= n:
                break
            # enclosure at every index on the way down
            pu, pl = bound, -bound
            for i in range(start, stop - 1, -1):
                pu = Z * (pu - c[i]); pl = Z * (pl - c[i])
                assert min(pu, pl) <= v[i] <= max(pu, pl), (name, K, a, i)
            if np.float64(pu).tobytes() == np.float64(pl).tobytes():
                assert np.float64(pu).tobytes() == np.float64(v[stop]).tobytes(), (name, K, a)
                met += K >= 52
            else:
                failed += K >= 52
    if name in ("noise_u16", "dim_u16", "beads", "unit_f32", "signed", "constant"):
        assert failed == 0 and met > 0, (name, met, failed)       # K >= 52: every tile certified on ordinary data
    if name == "zero_stretch":
        assert failed > 0                                          # exact zeros over more than K samples: fallback


def test_causal_recursion_is_monotone_too():
    """fl(x*g + fl(z*c)) is non-increasing in c: a line could also be CUT for the causal pass (not used by the kernels,
    which march from the start of the line; recorded because the same argument carries)."""
    rng = np.random.RandomState(3)
    x = rng.randint(100, 5000, 400).astype(np.float64)
    c = causal(x, x[0] * GAIN)
    B = 1.001 * GAIN * 65535.0 / (1.0 + Z)
    for a in range(80, 400, 40):
        pu, pl = B, -B
        for i in range(a - 60, a + 1):
            pu = x[i] * GAIN + Z * pu
            pl = x[i] * GAIN + Z * pl
            assert min(pu, pl) <= c[i] <= max(pu, pl)
        assert pu == pl == c[a]
