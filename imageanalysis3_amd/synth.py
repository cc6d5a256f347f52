"""Bit-stable synthetic FOV generator (repo-owned; SURVEY.md §8d "Synthetic inputs").

Everything here is built from integer hashing (splitmix64) and IEEE-754
``+ - * /``, ``floor`` and ``ldexp`` only — no libm ``exp/log/cos`` — so the
same stack is produced bit-for-bit on any host (fixtures are regenerated from the generator's
arguments on the GPU box instead of being shipped).
Golden fixtures under ``tests/golden`` store only *outputs*; inputs are
regenerated from ``(shape, n, seed, layout)``.

Generator ``G(shape, n, seed)``:
  background  400 + 15 * IH8(v)       IH8 = centred/normalised sum of eight 16-bit
                                      uniforms drawn from two splitmix64 words per voxel
                                      (Irwin-Hall approximation of N(0,1), |x| < 4.9)
  spots       h * dexp(-(dz²/σz² + dx²/σx² + dy²/σy²)/2) inside a (±5, ±8, ±8) window
              around round(centre); σ = (1.35, 1.9, 1.9) = reference ``_sigma_zxy``
              (/root/reference/__init__.py:10); h ~ U(1500, 6000); centres uniform with
              margin (6, 12, 12); layout 'isolated' (min separation 12 px, dart
              throwing) or 'clustered' (200 territories, σ_territory 15 px).
  value(v)    float32( ((bg(v) + c_s1(v)) + c_s2(v)) + ... )   spots in index order, f64.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_G = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)

SIGMA_ZXY = (1.35, 1.9, 1.9)
WIN = (5, 8, 8)
# std of the sum of eight independent U{0..65535}: sqrt(8 * (65536**2 - 1) / 12)
_IH8_MEAN = 8 * 65535 / 2.0
_IH8_STD = float(np.sqrt(8.0 * (65536.0 ** 2 - 1.0) / 12.0))


def splitmix64(x):
    """splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = (np.asarray(x, dtype=np.uint64) + _G)
        z = (z ^ (z >> np.uint64(30))) * _C1
        z = (z ^ (z >> np.uint64(27))) * _C2
        z = z ^ (z >> np.uint64(31))
    return z


def _key(seed, stream):
    with np.errstate(over="ignore"):
        return splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))


def uniform01(seed, stream, idx):
    """Deterministic U[0,1) doubles for integer counters ``idx``."""
    k = _key(seed, stream)
    h = splitmix64(k ^ np.asarray(idx, dtype=np.uint64))
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _ih8(h1, h2):
    m = np.uint64(0xFFFF)
    s = np.zeros(h1.shape, dtype=np.int64)
    for sh in (0, 16, 32, 48):
        s += ((h1 >> np.uint64(sh)) & m).astype(np.int64)
        s += ((h2 >> np.uint64(sh)) & m).astype(np.int64)
    return (s.astype(np.float64) - _IH8_MEAN) / _IH8_STD


def background(shape, seed, bg=400.0, noise=15.0, z0=0, z1=None):
    """float64 background planes z0:z1 of a (Z,X,Y) stack."""
    Z, X, Y = (int(s) for s in shape)
    if z1 is None:
        z1 = Z
    k = _key(seed, 1)
    out = np.empty((z1 - z0, X, Y), dtype=np.float64)
    plane = np.arange(X * Y, dtype=np.uint64).reshape(X, Y)
    with np.errstate(over="ignore"):
        for z in range(z0, z1):
            idx = plane + np.uint64(z * X * Y)
            h1 = splitmix64(k ^ (idx * np.uint64(2)))
            h2 = splitmix64(k ^ (idx * np.uint64(2) + np.uint64(1)))
            out[z - z0] = bg + noise * _ih8(h1, h2)
    return out


# ---- deterministic exp for x <= 0 (pure IEEE ops, no libm: identical on every host) ----
_LOG2E = 1.4426950408889634
_LN2_HI = 6.93147180369123816490e-01
_LN2_LO = 1.90821492927058770002e-10
_EXP_C = [1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0,
          1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0, 1.0]


def dexp(x):
    """exp(x) for -700 < x <= 0 with a fixed operation order (≈1e-13 relative)."""
    x = np.asarray(x, dtype=np.float64)
    k = np.floor(x * _LOG2E + 0.5)
    r = (x - k * _LN2_HI) - k * _LN2_LO
    p = np.full(x.shape, _EXP_C[0])
    for c in _EXP_C[1:]:
        p = p * r
        p = p + c
    return np.ldexp(p, k.astype(np.int32))


def spot_table(shape, n, seed, layout="isolated", h_range=(1500.0, 6000.0),
               margin=(6, 12, 12), min_sep=12.0, n_territories=200, territory_sigma=15.0):
    """Deterministic spot centres (n,3) float64 [z,x,y] and heights (n,)."""
    shape = np.asarray(shape, dtype=np.float64)
    margin = np.asarray(margin, dtype=np.float64)
    lo, hi = margin, shape - 1.0 - margin
    heights = h_range[0] + (h_range[1] - h_range[0]) * uniform01(seed, 3, np.arange(n))
    centers = np.zeros((n, 3), dtype=np.float64)
    if layout == "isolated":
        cell = float(min_sep)
        grid = {}
        i = 0
        ctr = 0
        while i < n:
            u = uniform01(seed, 2, np.arange(3 * ctr, 3 * ctr + 3))
            ctr += 1
            if ctr > 200 * n + 1000:
                raise RuntimeError("cannot place %d isolated spots in %s" % (n, shape))
            c = lo + (hi - lo) * u
            key = tuple(np.floor(c / cell).astype(np.int64))
            ok = True
            for dz in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    for dy in (-1, 0, 1):
                        for j in grid.get((key[0] + dz, key[1] + dx, key[2] + dy), ()):
                            d = centers[j] - c
                            if d[0] * d[0] + d[1] * d[1] + d[2] * d[2] < min_sep * min_sep:
                                ok = False
            if ok:
                centers[i] = c
                grid.setdefault(key, []).append(i)
                i += 1
    elif layout == "clustered":
        nt = int(n_territories)
        tu = uniform01(seed, 4, np.arange(3 * nt)).reshape(nt, 3)
        tc = lo + (hi - lo) * tu
        which = np.floor(uniform01(seed, 5, np.arange(n)) * nt).astype(np.int64)
        u = uniform01(seed, 6, np.arange(6 * n)).reshape(n, 3, 2)
        # sum-of-uniform offsets (deterministic, no libm): ~N(0, territory_sigma²) per axis
        g = (u[:, :, 0] + u[:, :, 1] - 1.0) * np.sqrt(6.0)
        scale = np.array([territory_sigma * 0.25, territory_sigma, territory_sigma])
        centers = tc[which] + g * scale
        centers = np.minimum(np.maximum(centers, lo), hi)
    elif layout == "uniform":
        u = uniform01(seed, 2, np.arange(3 * n)).reshape(n, 3)
        centers = lo + (hi - lo) * u
    else:
        raise ValueError("unknown layout %r" % (layout,))
    return centers, heights


def add_spots(im64, centers, heights, sigma=SIGMA_ZXY, z0=0):
    """im64[z - z0] += h*dexp(...) for every spot window, spots in index order."""
    Zc, X, Y = im64.shape
    inv = [1.0 / (s * s) for s in sigma]
    for c, h in zip(np.asarray(centers, dtype=np.float64), np.asarray(heights, dtype=np.float64)):
        r = np.floor(c + 0.5).astype(np.int64)
        a0, b0 = max(r[0] - WIN[0], z0), min(r[0] + WIN[0], z0 + Zc - 1)
        a1, b1 = max(r[1] - WIN[1], 0), min(r[1] + WIN[1], X - 1)
        a2, b2 = max(r[2] - WIN[2], 0), min(r[2] + WIN[2], Y - 1)
        if a0 > b0 or a1 > b1 or a2 > b2:
            continue
        dz = np.arange(a0, b0 + 1, dtype=np.float64) - c[0]
        dx = np.arange(a1, b1 + 1, dtype=np.float64) - c[1]
        dy = np.arange(a2, b2 + 1, dtype=np.float64) - c[2]
        q = ((dz * dz) * inv[0])[:, None, None] + ((dx * dx) * inv[1])[None, :, None]
        q = q + ((dy * dy) * inv[2])[None, None, :]
        im64[a0 - z0:b0 + 1 - z0, a1:b1 + 1, a2:b2 + 1] += h * dexp(-0.5 * q)
    return im64


def render(shape, centers, heights, seed, sigma=SIGMA_ZXY, bg=400.0, noise=15.0,
           dtype=np.float32):
    im64 = background(shape, seed, bg=bg, noise=noise)
    add_spots(im64, centers, heights, sigma=sigma)
    return quantise(im64, dtype)


def quantise(im64, dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return im64.astype(np.float32)
    if dtype == np.uint16:
        return np.floor(np.minimum(np.maximum(im64, 0.0), 65535.0) + 0.5).astype(np.uint16)
    if dtype == np.float64:
        return im64
    raise TypeError("unsupported dtype %s" % dtype)


def make_fov(shape, n_spots, seed, layout="isolated", dtype=np.float32, **kw):
    """Return (stack, centres, heights) for generator G(shape, n, seed)."""
    centers, heights = spot_table(shape, n_spots, seed, layout=layout,
                                  **{k: v for k, v in kw.items()
                                     if k in ("h_range", "margin", "min_sep", "n_territories",
                                              "territory_sigma")})
    im = render(shape, centers, heights, seed, dtype=dtype,
                **{k: v for k, v in kw.items() if k in ("sigma", "bg", "noise")})
    return im, centers, heights


def make_bead_pair(shape, n_beads, seed, drift, dtype=np.float32, h_range=(3000.0, 8000.0),
                   margin=(8, 24, 24), min_sep=16.0, noise=15.0):
    """Reference bead stack and a source stack whose beads sit at ``c + d``.

    With beads injected at ``c + d`` in the source, the drift returned by
    ``align_image`` is ≈ ``-d`` (SURVEY.md §3.4 sign convention).
    """
    centers, heights = spot_table(shape, n_beads, seed, layout="isolated", h_range=h_range,
                                  margin=margin, min_sep=min_sep)
    ref = render(shape, centers, heights, seed, dtype=dtype, noise=noise)
    src = render(shape, centers + np.asarray(drift, dtype=np.float64)[None, :], heights,
                 seed + 7919, dtype=dtype, noise=noise)
    return ref, src, centers, heights
