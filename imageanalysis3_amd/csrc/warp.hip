// 3-D warp by drift + optional per-voxel displacement field (reference: correction_tools/translate.py:5-31
// warp_3d_image and its production twins io_tools/load.py:438-453, classes/preprocess.py:918-946):
//   coords = grid (+ chromatic_profile) - drift ;  out = scipy.ndimage.map_coordinates(image, coords, order, mode)
//
// scipy semantics restated (verified bit-for-bit on the host against SciPy 1.15, see tests):
//   order 0            the sample at floor(c + 0.5) per axis, index clamped; 'constant': cval as for order 1
//   order 1            no prefilter; value = sum over the 2x2x2 corners, in C order, of ((c*w0)*w1)*w2 with
//                      w = [1-y, y], y = cc - floor(cc).  mode 'constant': cc < 0 or cc > n-1 on any axis -> cval.
//                      mode 'nearest': the coordinate is NOT clamped, out-of-range corner indices are.
//   order 3 'nearest'  input edge-padded by 12, cubic B-spline prefilter along axes 0,1,2 in float64
//                      (pole z = sqrt(3)-2 correctly rounded, gain (1-z)(1-1/z), half-sample-symmetric causal
//                      initialisation, anticausal c[n-1] *= z/(z-1)), then the 4x4x4 weighted sum in C order with
//                      weights w1=(y²(y-2)·3+4)/6, w2=(z²(z-2)·3+4)/6, w0=z³/6, w3=1-w0-w1-w2 (z=1-y).
//   order 3 'constant' not padded; prefilter with the whole-sample-symmetric boundary; cval outside [0, n-1]; taps
//                      mirrored about the first / last sample (section "order 3, mode 'constant'" below)
//   outputs            float32: cast; uint16: floor(t+0.5) clamped to [0, 65535].
// The coordinate grid (5 GB of float64 per FOV in the reference) is never materialised.
// Compiled with -ffp-contract=off.  HBM-bound streaming passes + an L2-served 64-tap gather.
#include "ia3_rt.h"
#include <math.h>
#include <stdio.h>

using namespace ia3rt;

#include "warp_iir0_kernel.inc"
using namespace ia3warpk;

namespace {

// stack depths with an axis-0 pass of their own (the depths the column kernels of the filters are built for)
#ifdef IA3_FOLD_DEPTHS
#define IA3_WARP_DEPTHS(X) IA3_FOLD_DEPTHS(X)
#else
#define IA3_WARP_DEPTHS(X) X(25) X(30) X(33) X(35) X(40) X(45) X(50) X(60)
#endif
#define IA3_POLE3 (-0.26794919243112270647)

template <class T> __device__ __forceinline__ T out_cvt(double t);
template <> __device__ __forceinline__ float out_cvt<float>(double t) { return (float)t; }
template <> __device__ __forceinline__ uint16_t out_cvt<uint16_t>(double t) {
  t = t > 0 ? t + 0.5 : 0.0;
  t = t > 65535.0 ? 65535.0 : t;
  return (uint16_t)(int)t;
}


// x / 6.0, correctly rounded, without the division sequence (15-20 dependent instructions, nine of them per voxel in
// the cubic weights): q = RN(x * RN(1/6)) is within an ulp of the quotient, the remainder r = x - 6q is exact in one
// fused multiply-add, and RN(q + r * RN(1/6)) is then the correctly rounded quotient (Markstein's theorem; the weights
// are far from the overflow / underflow ranges where it needs help).
__device__ __forceinline__ double div6(double x) {
  const double y = 0x1.5555555555555p-3;
  const double q = x * y;
  const double r = __builtin_fma(-6.0, q, x);
  return __builtin_fma(r, y, q);
}

// P[z,x,y] = im[clamp(z-12), clamp(x-12), clamp(y-12)] as float64 (np.pad(mode='edge'))
template <class T>
__global__ void spline_pad_k(const T* __restrict__ im, int Z, int X, int Y, double* __restrict__ P) {
  const int Xp = X + 2 * NPAD, Yp = Y + 2 * NPAD;
  const int y = blockIdx.x * 256 + threadIdx.x;
  if (y >= Yp) return;
  const int x = blockIdx.y, z = blockIdx.z;
  const int sz = clampi(z - NPAD, Z), sx = clampi(x - NPAD, X), sy = clampi(y - NPAD, Y);
  P[((size_t)z * Xp + x) * Yp + y] = (double)im[((size_t)sz * X + sx) * Y + sy];
}

// start-of-line value of the causal recursion for the 'nearest'/'reflect' boundary (see header)
// The sum runs over the whole line in SciPy.  When z^n underflows to zero (n >= 566) the mirror terms vanish exactly and
// the sum is cut where the rest provably cannot change it: every remaining term is at most |z|^i * bound in magnitude
// (bound = gain * largest |sample| the pass can meet), and an addend below a quarter ulp of the running sum leaves it
// unchanged under round-to-nearest, so once |sum| * 2^-55 > |z|^i * bound all further additions are no-ops.  The test
// is made after 64 terms and every 64 terms from there; a line whose leading samples are zero simply reads on.  The
// largest sample is 65535 for uint16 sources and is measured for float32 ones (absmax_f32_k); each pass of the
// prefilter can raise it by at most a factor 3 (the absolute sum of its impulse response).

__device__ __forceinline__ double iir_bound(const IirInit& q) {
  return q.amax_bits ? q.bound * (double)__uint_as_float(*q.amax_bits) : q.bound;
}
// true when no later term of the start sum can change `s` (see IirInit); NaN / inf bounds never pass
__device__ __forceinline__ bool iir_sum_settled(double s, double zi, double bound) {
  return fabs(s) * 0x1p-55 > fabs(zi) * bound || (zi == 0.0 && bound < INFINITY);
}

// bits of max |im| over a float32 stack: |x| as an unsigned integer orders finite < inf < NaN
__global__ __launch_bounds__(256) void absmax_f32_k(const float* __restrict__ im, size_t n, unsigned* __restrict__ out) {
  unsigned m = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const unsigned b = __float_as_uint(im[i]) & 0x7fffffffu;
    m = b > m ? b : m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned v = (unsigned)__shfl_xor((int)m, o); m = v > m ? v : m; }
  if ((threadIdx.x & 63) == 0 && m > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, m);
}

// start value of the causal recursion of a strided line (element i at c + i*stride), see IirInit
__device__ __forceinline__ double iir_start_strided(const double* __restrict__ c, size_t stride, int n, const IirInit& q) {
  const double z = q.z, g = q.gain;
  const double c0 = c[0] * g;
  double s;
  if (q.full) {
    s = c0 + q.zn * (c[(size_t)(n - 1) * stride] * g);
    double zi = z;
    for (int i = 1; i < n; ++i) {
      s += zi * (c[(size_t)i * stride] * g + q.zn * (c[(size_t)(n - 1 - i) * stride] * g));
      zi *= z;
    }
  } else {
    const double bound = iir_bound(q);
    s = c0;
    double zi = z;
    for (int i = 1; i < n;) {
      const int e = i + 63 < n ? i + 63 : n;
      for (; i < e; ++i) { s += zi * (c[(size_t)i * stride] * g); zi *= z; }
      if (iir_sum_settled(s, zi, bound)) break;
    }
  }
  s *= q.scale;
  s += c0;
  return s;
}

// Two sweeps over a strided line, in place, from sample `from` (whose causal value `first` is known) to the end and
// back: [from, n) holds raw samples on entry and coefficients on exit.
// The recursions are serial in `prev`, their loads are not: eight samples are fetched ahead of the eight dependent
// updates, so a thread keeps eight loads in flight instead of one (the kernel ran at 1.6 TB/s, latency-bound).
// MIRROR: the anticausal start of the whole-sample-symmetric boundary (mode 'constant', below) instead of the
// half-sample-symmetric one; needs the causal values of the last TWO samples (from <= n - 2).
template <bool MIRROR = false>
__device__ __forceinline__ void iir_two_sweeps_strided(double* __restrict__ c, size_t stride, int from, int n, double first,
                                                       double z, double g) {
  constexpr int B = 8;   // 16 in flight: no faster (2.36 against 2.25 ms)
  double prev = first;
  c[(size_t)from * stride] = prev;
  int i = from + 1;
  for (; i + B <= n; i += B) {
    double in[B];
#pragma unroll
    for (int k = 0; k < B; ++k) in[k] = c[(size_t)(i + k) * stride];
#pragma unroll
    for (int k = 0; k < B; ++k) {
      const double v = in[k] * g + z * prev;
      c[(size_t)(i + k) * stride] = v;
      prev = v;
    }
  }
  for (; i < n; ++i) {
    double v = c[(size_t)i * stride] * g + z * prev;
    c[(size_t)i * stride] = v;
    prev = v;
  }
  if (MIRROR) prev = ((z * c[(size_t)(n - 2) * stride] + prev) * z) / (z * z - 1.0);   // (c[n-2]: this thread's own store)
  else prev = prev * (z / (z - 1.0));
  c[(size_t)(n - 1) * stride] = prev;
  i = n - 2;
  for (; i - (B - 1) >= from; i -= B) {
    double in[B];
#pragma unroll
    for (int k = 0; k < B; ++k) in[k] = c[(size_t)(i - k) * stride];
#pragma unroll
    for (int k = 0; k < B; ++k) {
      const double v = z * (prev - in[k]);
      c[(size_t)(i - k) * stride] = v;
      prev = v;
    }
  }
  for (; i >= from; --i) {
    double v = z * (prev - c[(size_t)i * stride]);
    c[(size_t)i * stride] = v;
    prev = v;
  }
}

// IIR along a strided axis: line p (lane along the contiguous axis), element i at base + i*stride
__global__ __launch_bounds__(256) void spline_iir_strided_k(double* __restrict__ P, int inner, size_t stride, int n,
                                                            size_t outer_stride, IirInit q, const int* __restrict__ zr) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= inner) return;
  if (zr && ((int)blockIdx.y < zr[0] || (int)blockIdx.y > zr[1])) return;   // (blockIdx.y = plane: not read by the gather)
  double* c = P + (size_t)blockIdx.y * outer_stride + p;
  iir_two_sweeps_strided(c, stride, 0, n, iir_start_strided(c, stride, n, q), q.z, q.gain);
}

// The same pass with every sample read once and written once (the two sweeps above move the float64 coefficient stack
// through HBM twice).  The anticausal recursion  v[i] = z * (v[i+1] - c[i])  of a tile [a, a+T) needs v[a+T], which the
// two-sweep form knows only after it has been to the end of the line.  Here it is CERTIFIED from the K causal values
// behind the tile: two chains run the same recursion down from a+T+K-1, one started at +bound, one at -bound, where
// |v| <= bound on the whole line (IirInit: v is at most three times the largest sample of the pass).  Rounded
// subtraction is non-decreasing in v[i+1] and the rounded product with z < 0 non-increasing, so by induction the true
// value lies between the two chains at every index; their distance shrinks by |z| = 0.268 per sample, and once both
// hold the same bit pattern the true value has that bit pattern too.  The distance falls below an ulp of a value v
// after log(bound / ulp(v)) / log(1 / 0.268) steps — 33 for bound / |v| = 1000, five more per decade — and then the
// chains coincide or sit on neighbouring doubles and merge with probability ~0.73 per further step.  With K = 52 they
// agree unless |v| is six orders of magnitude below the bound (exact zeros: dark borders) or, with probability
// ~1e-7 per tile, keep straddling rounding boundaries; K = 32 fails on most lines of a noise image, K = 48 is at the edge
// for a dim uint16 background under the 65535 bound.  A thread whose chains disagree finishes ITS line with the two
// sweeps from where it stands (the rest of that line is still raw).  The causal recursion needs no certificate: the
// thread marches along the line.  The window of T + K causal values lives in registers.
template <int T, int K, int OCC>
__global__ __launch_bounds__(64, OCC) void spline_iir_strided_1p_k(double* __restrict__ P, int inner, size_t stride, int n,
                                                                   size_t outer_stride, IirInit q, int warm, const int* __restrict__ zr) {
  constexpr int W = T + K;
  const int p = blockIdx.x * 64 + threadIdx.x;
  if (p >= inner) return;
  if ((int)blockIdx.y < zr[0] || (int)blockIdx.y > zr[1]) return;   // a plane the gather will not read
  double* c = P + (size_t)blockIdx.y * outer_stride + p;
  const double z = q.z, g = q.gain;
  const double bound = iir_bound(q);
  double w[W];                       // causal values of [a, a + W)
  w[0] = iir_start_strided(c, stride, n, q);
#pragma unroll
  for (int j = 1; j < W; ++j) w[j] = c[(size_t)j * stride];          // n >= 2 W (host)
#pragma unroll
  for (int j = 1; j < W; ++j) w[j] = w[j] * g + z * w[j - 1];
  int a = 0;
  bool slow = false;
  while (a + W < n) {
    double pu = bound, pl = -bound;   // stand for v[a + W]
#pragma unroll
    for (int j = W - 1; j >= T; --j)
      if (j - T < warm) { pu = z * (pu - w[j]); pl = z * (pl - w[j]); }
    if (__double_as_longlong(pu) != __double_as_longlong(pl)) { slow = true; break; }
    double v = pu;
    double* ca = c + (size_t)a * stride;
#pragma unroll
    for (int j = T - 1; j >= 0; --j) { v = z * (v - w[j]); ca[(size_t)j * stride] = v; }
#pragma unroll
    for (int j = 0; j < K; ++j) w[j] = w[j + T];
    const int e = a + W;
    const double* ce = c + (size_t)e * stride;
    // (fetching these a step ahead into registers of their own was measured: 1.37 - 1.40 ms at T = 8 / 10 against 1.34 for
    // this form at T = 12, which then spills; three waves per SIMD already keep loads in flight while one computes)
    if (e + T <= n) {
#pragma unroll
      for (int j = 0; j < T; ++j) w[K + j] = ce[(size_t)j * stride];
#pragma unroll
      for (int j = 0; j < T; ++j) w[K + j] = w[K + j] * g + z * w[K + j - 1];
    } else {
#pragma unroll
      for (int j = 0; j < T; ++j) if (e + j < n) w[K + j] = ce[(size_t)j * stride];
#pragma unroll
      for (int j = 0; j < T; ++j) if (e + j < n) w[K + j] = w[K + j] * g + z * w[K + j - 1];
    }
    a += T;
  }
  if (!slow) {      // the window reaches the end of the line: K < n - a <= W samples, exact from the last one down
    const int m = n - a;
    double* ca = c + (size_t)a * stride;
    double v = 0.0;
#pragma unroll
    for (int j = W - 1; j >= 0; --j) {
      if (j == m - 1) { v = w[j] * (z / (z - 1.0)); ca[(size_t)j * stride] = v; }
      else if (j < m - 1) { v = z * (v - w[j]); ca[(size_t)j * stride] = v; }
    }
  } else {          // [0, a) is final, [a, n) raw, w[0] the causal value of sample a
    iir_two_sweeps_strided(c, stride, a, n, w[0], z, g);
  }
}

// Padding and the axis-0 recursion in one pass: a padded z line (<= ZMAX samples) lives in registers, so the padded
// float64 volume is written once instead of written, read, written, read and written again.  Same operations in the
// same order as spline_pad_k followed by spline_iir_strided_k with the faithful (full) start sum.
template <class T, int ZMAX>
__global__ __launch_bounds__(256, 2) void spline_pad_iir0_k(const T* __restrict__ im, int Z, int X, int Y, double* __restrict__ P,
                                                         IirInit q, const int* __restrict__ zr) {
  const int Xp = X + 2 * NPAD, Yp = Y + 2 * NPAD, n = Z + 2 * NPAD;
  const size_t plane = (size_t)Xp * Yp;
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= plane) return;
  const int x = (int)(p / Yp), y = (int)(p - (size_t)x * Yp);
  const size_t src = (size_t)clampi(x - NPAD, X) * Y + clampi(y - NPAD, Y);
  const double z = q.z, g = q.gain;
  const size_t zs = (size_t)X * Y;
  auto ld = [&](int i) { return (double)im[(size_t)clampi(i - NPAD, Z) * zs + src] * g; };
  // start value of the causal recursion: the faithful sum over the whole line, streamed from the source
  const double c0 = ld(0);
  double s = c0 + q.zn * ld(n - 1), zi = z;
  for (int i = 1; i < n; ++i) {
    s += zi * (ld(i) + q.zn * ld(n - 1 - i));
    zi *= z;
  }
  double c[ZMAX];
#pragma unroll
  for (int i = 0; i < ZMAX; ++i) c[i] = i < n ? ld(i) : 0.0;
  s *= q.scale;
  s += c0;
  double prev = s;
  c[0] = prev;
#pragma unroll
  for (int i = 1; i < ZMAX; ++i) if (i < n) { const double v = c[i] + z * prev; c[i] = v; prev = v; }
  prev = prev * (z / (z - 1.0));
#pragma unroll
  for (int i = ZMAX - 1; i >= 0; --i) {
    if (i == n - 1) c[i] = prev;
    else if (i < n - 1) { const double v = z * (prev - c[i]); c[i] = v; prev = v; }
  }
  const int p_lo = zr[0], p_hi = zr[1];   // (planes the gather will not read are not stored, see zrange_k)
#pragma unroll
  for (int i = 0; i < ZMAX; ++i) if (i < n && i >= p_lo && i <= p_hi) P[(size_t)i * plane + p] = c[i];
}

// IIR along the contiguous axis: every wave owns LPW lines (lanes 0..LPW-1 run the recursions) and marches them in
// TW-element tiles that are transposed through wave-private LDS tiles.  One wave instruction moves 64/TW row pieces of
// TW doubles.  Measured on 74 x 2072 x 2072 (10 GB moved by the two-sweep form): 64 lines x 16 samples 3.10 ms, the same
// with the chain in registers 2.9, 16 lines x 64 samples 2.5, 8 lines x 64 samples at four waves per SIMD 2.3 (4.3 TB/s
// of mixed reads and writes): the pass wants many small waves with whole 512-byte pieces of a line per access, not
// wide tiles.
// warm > 0 and at least three tiles per line: ONE pass (see spline_iir_strided_1p_k for the argument): a tile gets its
// causal recursion, the two bounding chains run back through it (a whole tile: 64 samples), and when they meet in every
// line of the wave the tile before it can have its exact anticausal recursion and be stored — each sample is read once
// and written once.  If the chains of any line disagree the wave finishes its lines with the two sweeps from the
// first tile that is still raw in memory.
template <int TW, int LPW, int OCC>
__global__ __launch_bounds__(256, OCC) void spline_iir_contig_k(double* __restrict__ P, size_t n_lines, int n, IirInit q, int warm,
                                                                const int* __restrict__ zr, int lines_per_plane) {
  constexpr int RPI = 64 / TW;   // rows moved per wave instruction
  constexpr int NR = LPW / RPI;  // wave instructions per tile
  constexpr int SC = 16;         // samples of a line in registers at a time
  static_assert(64 % TW == 0 && LPW % RPI == 0 && LPW <= 64 && TW % SC == 0, "tile shape");
  __shared__ double tiles[4][3][LPW][TW + 1];
  const int wv = threadIdx.x >> 6, t = threadIdx.x & 63;
  const size_t l0 = ((size_t)blockIdx.x * 4 + wv) * LPW;
  if (l0 >= n_lines) return;                 // whole wave
  {   // all of this wave's lines in planes the gather will not read (zrange_k): nothing to do
    const size_t l1 = l0 + LPW - 1 < n_lines ? l0 + LPW - 1 : n_lines - 1;
    const int pa = (int)(l0 / (size_t)lines_per_plane), pb = (int)(l1 / (size_t)lines_per_plane);
    if (pb < zr[0] || pa > zr[1]) return;
  }
  const bool mine = t < LPW && l0 + t < n_lines;
  const int lt = t < LPW ? t : 0;            // row of the tile this lane's recursion runs on
  const int col = t % TW, rsub = t / TW;
  const double z = q.z, g = q.gain;
  // Tiles on their way from HBM while the current one is processed.  Raw values: the gain is applied when a tile is
  // put into LDS (a multiplication at the load would make the wave wait for the load it has just issued).  The pass is
  // bound by the bytes it keeps in flight: at ~5 us of loaded memory latency, one 4 KB tile per wave and 12 waves per
  // CU are 2.5 TB/s; DEPTH tiles ahead lift that above what HBM delivers.
  constexpr int DEPTH = 3;
  double pre[DEPTH][NR];
  double pre_scale = 1.0;
  auto fetch_to = [&](int d, int y0, int w) {
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int r = j * RPI + rsub;
      pre[d][j] = (l0 + r < n_lines && col < w) ? P[(l0 + r) * (size_t)n + y0 + col] : 0.0;
    }
  };
  auto fetch = [&](int y0, int w, double scale) { fetch_to(0, y0, w); pre_scale = scale; };
  auto stash = [&](int b) {
#pragma unroll
    for (int j = 0; j < NR; ++j) tiles[wv][b][j * RPI + rsub][col] = pre[0][j] * pre_scale;
  };
  auto shift_pre = [&]() {
#pragma unroll
    for (int d = 0; d + 1 < DEPTH; ++d)
#pragma unroll
      for (int j = 0; j < NR; ++j) pre[d][j] = pre[d + 1][j];
  };
  auto store_tile = [&](int b, int y0, int w) {
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int r = j * RPI + rsub;
      if (l0 + r < n_lines && col < w) P[(l0 + r) * (size_t)n + y0 + col] = tiles[wv][b][r][col];
    }
  };
  // causal recursion over row[i0 .. w) continuing from prev; whole tiles keep SC samples in registers: one dependent
  // chain, no LDS in it
  auto causal_row = [&](int bf, int i0, int w, double prev) {
    double* row = &tiles[wv][bf][lt][0];
    if (i0 == 0 && w == TW) {
#pragma unroll 1
      for (int b = 0; b < TW; b += SC) {
        double r[SC];
#pragma unroll
        for (int i = 0; i < SC; ++i) r[i] = row[b + i];
#pragma unroll
        for (int i = 0; i < SC; ++i) { const double v = r[i] + z * prev; r[i] = v; prev = v; }
#pragma unroll
        for (int i = 0; i < SC; ++i) row[b + i] = r[i];
      }
    } else {
      for (int i = i0; i < w; ++i) { const double v = row[i] + z * prev; row[i] = v; prev = v; }
    }
    return prev;
  };
  // anticausal recursion over row[w-1 .. 0] continuing from v (the value behind the tile)
  auto anti_row = [&](int bf, int w, double v) {
    double* row = &tiles[wv][bf][lt][0];
    if (w == TW) {
#pragma unroll 1
      for (int b = TW - SC; b >= 0; b -= SC) {
        double r[SC];
#pragma unroll
        for (int i = 0; i < SC; ++i) r[i] = row[b + i];
#pragma unroll
        for (int i = SC - 1; i >= 0; --i) { const double u = z * (v - r[i]); r[i] = u; v = u; }
#pragma unroll
        for (int i = 0; i < SC; ++i) row[b + i] = r[i];
      }
    } else {
      for (int i = w - 1; i >= 0; --i) { const double u = z * (v - row[i]); row[i] = u; v = u; }
    }
    return v;
  };
  const int ntile = (n + TW - 1) / TW;
  // start value of the causal recursion (see make_init)
  double prev = 0.0;
  if (q.full) {   // short lines: the sum needs the line from both ends, read it directly
    if (mine) {
      const double* line = P + (l0 + t) * (size_t)n;
      const double c0 = line[0] * g;
      double zi = z;
      double s = c0 + q.zn * (line[n - 1] * g);
      for (int i = 1; i < n; ++i) { s += zi * (line[i] * g + q.zn * (line[n - 1 - i] * g)); zi *= z; }
      s *= q.scale;
      s += c0;
      prev = s;
    }
  } else {        // until the sum is settled (IirInit), through coalesced tiles (a lane reading its own line touches 64 cache lines per load)
    static_assert(64 % TW == 0, "the settled test is made at multiples of 64 terms");
    const double bound = iir_bound(q);
    double s = 0.0, c0 = 0.0, zi = z;
    bool open = mine;   // this lane's sum can still change
    for (int y0 = 0; y0 < n; y0 += TW) {
      const int w = n - y0 < TW ? n - y0 : TW;
      fetch(y0, w, g);
      stash(0);
      __builtin_amdgcn_wave_barrier();
      if (open) {
        int i0 = 0;
        if (y0 == 0) { c0 = tiles[wv][0][lt][0]; s = c0; i0 = 1; }
        for (int i = i0; i < w; ++i) { s += zi * tiles[wv][0][lt][i]; zi *= z; }
        if ((y0 + TW) % 64 == 0 && iir_sum_settled(s, zi, bound)) open = false;
      }
      __builtin_amdgcn_wave_barrier();
      if (!__any(open)) break;
    }
    s *= q.scale;
    s += c0;
    prev = s;
  }
  int k0 = 0;                    // the two sweeps below start at this tile, with `prev` the causal value entering it
  if (warm > 0 && ntile >= 3) {
    // Lanes 0..LPW-1 own the lines (group 0).  The backward recursion of the certificate and the backward recursion
    // of a tile's exact pass are the same operation on different rows, so they share one instruction stream: in the
    // backward pass of step j, group 0 runs the chain from +bound and group 1 (lanes LPW..2 LPW-1) the chain from
    // -bound through tile j, while group 2 runs the exact anticausal pass of tile j-2 from the value step j-1
    // certified.  Three tiles are resident per wave; a full tile costs one forward and one backward pass of 64 steps.
    static_assert(3 * LPW <= 64, "three lane groups");
    const double bound = iir_bound(q);
    const int L = ntile - 1, wlast = n - L * TW;
    const int grp = t / LPW, ln = t % LPW;
    const bool lane_on = grp < 3 && l0 + ln < n_lines;
    const int wsteps = ((warm < TW ? warm : TW) + SC - 1) / SC * SC;   // warm-up in whole register chunks
    // backward recursion over the whole tile in buffer bf, row ln; chains of groups 0 / 1 (re)start from `init` at
    // sample wsteps - 1 (the shortened warm-ups of the tests); wr: the lane keeps its results
    auto back_pass = [&](int bf, double v, double init, bool chain, bool wr) {
      double* row = &tiles[wv][bf][ln][0];
#pragma unroll 1
      for (int b = TW - SC; b >= 0; b -= SC) {
        if (chain && b + SC == wsteps) v = init;
        double r[SC];
#pragma unroll
        for (int i = 0; i < SC; ++i) r[i] = row[b + i];
#pragma unroll
        for (int i = SC - 1; i >= 0; --i) { const double u = z * (v - r[i]); r[i] = u; v = u; }
        if (wr) {
#pragma unroll
          for (int i = 0; i < SC; ++i) row[b + i] = r[i];
        }
      }
      return v;
    };
    fetch(0, TW, g);
    stash(0);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)       // tiles 1 .. DEPTH
      if (1 + d <= L) fetch_to(d, (1 + d) * TW, 1 + d == L ? wlast : TW);
    double cin_prev = prev, cin_cur = prev;   // causal values entering tiles j - 1 and j (tile 0: the start value itself)
    double cprev = prev;
    if (mine) { tiles[wv][0][lt][0] = prev; cprev = causal_row(0, 1, TW, prev); }
    double vcert = 0.0;                       // group 2: the value behind the tile it finishes next
    bool failed = false;
    for (int j = 1; j < L; ++j) {             // full tiles; pre: raw tile j
      const int bj = j % 3, bjm2 = (j + 1) % 3;
      stash(bj);
      __builtin_amdgcn_wave_barrier();
      shift_pre();
      if (j + DEPTH <= L) fetch_to(DEPTH - 1, (j + DEPTH) * TW, j + DEPTH == L ? wlast : TW);
      cin_prev = cin_cur;
      cin_cur = cprev;
      if (mine) cprev = causal_row(bj, 0, TW, cprev);
      __builtin_amdgcn_wave_barrier();
      double v = grp == 0 ? bound : (grp == 1 ? -bound : vcert);
      const bool fin = grp == 2 && j >= 2;    // tile j - 2 gets its exact pass
      if (lane_on && (grp < 2 || fin)) v = back_pass(grp == 2 ? bjm2 : bj, v, v, grp < 2, fin);
      const double vl = __shfl(v, (t + LPW) & 63);          // group 0 looks at group 1's chain
      const double vu = __shfl(v, (t + 64 - 2 * LPW) & 63); // group 2 takes group 0's
      const bool fail = mine && __double_as_longlong(v) != __double_as_longlong(vl);
      vcert = vu;
      __builtin_amdgcn_wave_barrier();
      if (j >= 2) store_tile(bjm2, (j - 2) * TW, TW);
      __builtin_amdgcn_wave_barrier();
      if (__any(fail)) { failed = true; k0 = j - 1; prev = cin_prev; break; }
    }
    if (!failed) {
      // the last tile (wlast samples) ends the line: exact from its last sample; tiles L-1 (from the value the last tile
      // hands down, group 0) and L-2 (from the certified value, group 2) follow in one backward pass
      const int bL = L % 3, bLm1 = (L + 2) % 3, bLm2 = (L + 1) % 3;
      stash(bL);
      __builtin_amdgcn_wave_barrier();
      double v = vcert;
      if (mine) {
        causal_row(bL, 0, wlast, cprev);
        v = tiles[wv][bL][lt][wlast - 1] * (z / (z - 1.0));
        tiles[wv][bL][lt][wlast - 1] = v;
        v = anti_row(bL, wlast - 1, v);
      }
      __builtin_amdgcn_wave_barrier();
      store_tile(bL, L * TW, wlast);
      const bool fin2 = grp == 2 && L >= 2;
      if (lane_on && (grp == 0 || fin2)) back_pass(grp == 2 ? bLm2 : bLm1, v, v, false, true);
      __builtin_amdgcn_wave_barrier();
      store_tile(bLm1, (L - 1) * TW, TW);
      store_tile(bLm2, (L - 2) * TW, TW);
      return;
    }
    __builtin_amdgcn_wave_barrier();
  }
  // two sweeps from tile k0 on: [k0 * TW, n) is raw
  fetch(k0 * TW, n - k0 * TW < TW ? n - k0 * TW : TW, g);
  for (int k = k0; k < ntile; ++k) {  // causal
    const int y0 = k * TW, w = n - y0 < TW ? n - y0 : TW;
    stash(0);
    __builtin_amdgcn_wave_barrier();
    if (k + 1 < ntile) fetch(y0 + TW, n - y0 - TW < TW ? n - y0 - TW : TW, g);
    if (mine) {
      if (k == 0) { tiles[wv][0][lt][0] = prev; prev = causal_row(0, 1, w, prev); }
      else prev = causal_row(0, 0, w, prev);
    }
    __builtin_amdgcn_wave_barrier();
    store_tile(0, y0, w);
    __builtin_amdgcn_wave_barrier();
  }
  {
    const int y0 = (ntile - 1) * TW;
    fetch(y0, n - y0, 1.0);   // written just above by this wave: program order, same addresses
  }
  for (int k = ntile - 1; k >= k0; --k) {  // anticausal
    const int y0 = k * TW, w = n - y0 < TW ? n - y0 : TW;
    stash(0);
    __builtin_amdgcn_wave_barrier();
    if (k > k0) fetch(y0 - TW, TW, 1.0);
    if (mine) {
      if (k == ntile - 1) {
        prev = tiles[wv][0][lt][w - 1] * (z / (z - 1.0));
        tiles[wv][0][lt][w - 1] = prev;
        prev = anti_row(0, w - 1, prev);
      } else prev = anti_row(0, w, prev);
    }
    __builtin_amdgcn_wave_barrier();
    store_tile(0, y0, w);
    __builtin_amdgcn_wave_barrier();
  }
}

__device__ __forceinline__ double field_at(const void* f, int fdt, size_t i) {
  return fdt == 1 ? (double)((const float*)f)[i] : ((const double*)f)[i];
}

// ---- the padded planes a cubic warp reads ------------------------------------------------------------------------------
// SciPy pads every axis by 12 samples and filters the padded volume; a stack of 50 planes becomes 74, and the passes along
// x and y and the stores of the z pass spend a third of their time on planes no tap ever reaches: output plane z reads the
// coefficient planes floor(z - drift_z + field_z + 12) - 1 ... + 2.  zrange_k turns the drift and the range of the field's
// z component (field_zminmax_k; nothing to scan without a field) into the first and last padded plane that can be read;
// the prefilter kernels take the two words from device memory — no host round trip — and
// leave the other planes alone.  The coefficients of the planes that are computed do not change (the z recursion runs over
// the whole padded line in registers; x and y lines lie inside a plane).
constexpr int ZR_BLOCKS = 2048;
// per block: [smallest value, largest value, 1 if a value was not finite] of fz[0 .. n)
template <class F>
__global__ __launch_bounds__(256) void field_zminmax_k(const F* __restrict__ fz, size_t n, float* __restrict__ part) {
  float lo = __builtin_huge_valf(), hi = -__builtin_huge_valf();
  bool bad = false;
  auto take = [&](F v) {
    bad = bad || !(v - v == (F)0);                         // NaN / inf: every plane is kept
    const float f = (float)v;                              // (float64 fields: rounded to nearest; zrange_k's margin covers half an ulp)
    lo = fminf(lo, f); hi = fmaxf(hi, f);
  };
  constexpr int PER = 16 / (int)sizeof(F);                 // values per 16-byte load
  typedef F vec __attribute__((ext_vector_type(PER)));
  const size_t nv = n / PER;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
    const vec q = ((const vec*)fz)[i];
#pragma unroll
    for (int k = 0; k < PER; ++k) take(q[k]);
  }
  if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n - nv * PER)) take(fz[nv * PER + threadIdx.x]);
  __shared__ float slo[256], shi[256];
  __shared__ int sbad[256];
  slo[threadIdx.x] = lo; shi[threadIdx.x] = hi; sbad[threadIdx.x] = bad ? 1 : 0;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) {
      slo[threadIdx.x] = fminf(slo[threadIdx.x], slo[threadIdx.x + k]);
      shi[threadIdx.x] = fmaxf(shi[threadIdx.x], shi[threadIdx.x + k]);
      sbad[threadIdx.x] |= sbad[threadIdx.x + k];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[3 * blockIdx.x] = slo[0]; part[3 * blockIdx.x + 1] = shi[0]; part[3 * blockIdx.x + 2] = sbad[0] ? 1.f : 0.f; }
}
// the partials of field_zminmax_k folded into one triple (kept beside a constant field, runtime.cpp: const_note)
__global__ __launch_bounds__(256) void zminmax_fold_k(const float* __restrict__ part, int nb, float* __restrict__ out) {
  __shared__ float slo[256], shi[256], sb[256];
  float lo = __builtin_huge_valf(), hi = -__builtin_huge_valf(), bad = 0.f;
  for (int b = threadIdx.x; b < nb; b += 256) { lo = fminf(lo, part[3 * b]); hi = fmaxf(hi, part[3 * b + 1]); bad = fmaxf(bad, part[3 * b + 2]); }
  slo[threadIdx.x] = lo; shi[threadIdx.x] = hi; sb[threadIdx.x] = bad;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) {
      slo[threadIdx.x] = fminf(slo[threadIdx.x], slo[threadIdx.x + k]);
      shi[threadIdx.x] = fmaxf(shi[threadIdx.x], shi[threadIdx.x + k]);
      sb[threadIdx.x] = fmaxf(sb[threadIdx.x], sb[threadIdx.x + k]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = slo[0]; out[1] = shi[0]; out[2] = sb[0]; }
}
__global__ void zrange_k(const float* __restrict__ part, int nb, double dz, int Z, int* __restrict__ zr) {
  double lo = 0.0, hi = 0.0;
  bool bad = false;
  for (int b = 0; b < nb; ++b) {
    lo = b == 0 ? (double)part[0] : fmin(lo, (double)part[3 * b]);
    hi = b == 0 ? (double)part[1] : fmax(hi, (double)part[3 * b + 1]);
    bad = bad || part[3 * b + 2] != 0.f;
  }
  const int Zp = Z + 2 * NPAD;
  // coordinates 0 - dz + lo ... (Z - 1) - dz + hi (the two orders of adding drift and field differ by rounding only: the
  // margin covers it); first tap floor(c + 12) - 1, last tap floor(c + 12) + 2
  // (the extremes of a float64 field were rounded to float, 6e-8 relative; the two orders of adding drift and field differ
  // by an ulp of the coordinate: a margin of 1e-4 + 1e-6 |value| voxels is orders of magnitude more than either)
  lo = lo - (1e-4 + 1e-6 * fabs(lo)); hi = hi + (1e-4 + 1e-6 * fabs(hi));
  const double cmin = -dz + lo + (double)NPAD, cmax = (double)(Z - 1) - dz + hi + (double)NPAD;
  int p_lo = 0, p_hi = Zp - 1;
  if (!bad && cmin - cmin == 0.0 && cmax - cmax == 0.0) {
    const double a = floor(cmin) - 1.0, b = floor(cmax) + 2.0;   // first tap floor(c) - 1, last tap floor(c) + 2
    p_lo = a < 0.0 ? 0 : (a > (double)(Zp - 1) ? Zp - 1 : (int)a);
    p_hi = b < 0.0 ? 0 : (b > (double)(Zp - 1) ? Zp - 1 : (int)b);
  }
  zr[0] = p_lo; zr[1] = p_hi;
}

// ---- order 3, mode 'constant' (translate.py:5-31 called with warp_order=3 and its default border mode) --------------
// SciPy's rules for this mode (restated in numpy and compared bit for bit, tests/test_host_logic_cpu.py): the input is
// NOT padded; the prefilter runs with the whole-sample-symmetric ("mirror") boundary —
//     c[0]   = (c[0] + z^(n-1) c[n-1] + sum_{i=1}^{n-2} z^i (c[i] + z^(n-1) c[n-1-i])) / (1 - z^(2n-2))      (same order)
//     c[n-1] = ((z c[n-2] + c[n-1]) z) / (z² - 1)
// — a coordinate below 0 or above n-1 on any axis gives cval, and taps that leave the array are mirrored about its
// first / last sample.  Off the production path (the twins in io_tools/load.py and classes/preprocess.py use 'nearest'):
// plain kernels, one thread per line with two sweeps, one output per thread.
struct MirInit {
  double z, gain, zn1, den;   // zn1 = z^(n-1), den = 1 - zn1²
  int n_sum;                  // terms of the start sum that can matter: z^i (by repeated multiplication) is 0 from there on
};
template <class T>
__global__ __launch_bounds__(256) void spline_cvt_k(const T* __restrict__ im, size_t n, double* __restrict__ P) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) P[i] = (double)im[i];
}
// line (o, p): element i at P + o * outer_stride + p * inner_stride + i * stride
__global__ __launch_bounds__(256) void spline_mirror_k(double* __restrict__ P, int inner, size_t inner_stride, size_t stride, int n,
                                                       size_t outer_stride, MirInit q) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= inner) return;
  double* c = P + (size_t)blockIdx.y * outer_stride + (size_t)p * inner_stride;
  const double z = q.z, g = q.gain;
  double s = c[0] * g + q.zn1 * (c[(size_t)(n - 1) * stride] * g);
  double zi = z;
  if (q.zn1 != 0.0) {
    for (int i = 1; i < n - 1; ++i) { s += zi * (c[(size_t)i * stride] * g + q.zn1 * (c[(size_t)(n - 1 - i) * stride] * g)); zi *= z; }
  } else {   // c[i] + 0 * c[n-1-i] = c[i]; beyond n_sum every term is a zero
    const int e = n - 1 < q.n_sum ? n - 1 : q.n_sum;
    for (int i = 1; i < e; ++i) { s += zi * (c[(size_t)i * stride] * g); zi *= z; }
  }
  s /= q.den;
  iir_two_sweeps_strided<true>(c, stride, 0, n, s, z, g);
}
__device__ __forceinline__ int mirror_idx(int i, int n) {   // SciPy's edge offsets: ... 2 1 | 0 1 2 ... n-1 | n-2 n-3 ...
  const int s2 = 2 * n - 2;
  if (i < 0) { i = s2 * (-i / s2) + i; i = i <= 1 - n ? i + s2 : -i; }
  else if (i >= n) { i -= s2 * (i / s2); if (i >= n) i = s2 - i; }
  return i;
}
template <class T>
__global__ __launch_bounds__(256) void warp_cubic_mirror_k(const double* __restrict__ C, int Z, int X, int Y, double dz, double dx,
                                                           double dy, const void* __restrict__ field, int fdt, double cval,
                                                           T* __restrict__ out) {
  const int y = blockIdx.x * 256 + threadIdx.x;
  if (y >= Y) return;
  const int x = blockIdx.y, zq = blockIdx.z;
  const size_t o = ((size_t)zq * X + x) * Y + y, V = (size_t)Z * X * Y;
  double cc[3] = {(double)zq, (double)x, (double)y};
  if (fdt & 16) {
    cc[0] = cc[0] - dz; cc[1] = cc[1] - dx; cc[2] = cc[2] - dy;
    if (field) { cc[0] = cc[0] + field_at(field, fdt & 3, o); cc[1] = cc[1] + field_at(field, fdt & 3, V + o); cc[2] = cc[2] + field_at(field, fdt & 3, 2 * V + o); }
  } else {
    if (field) { cc[0] = cc[0] + field_at(field, fdt & 3, o); cc[1] = cc[1] + field_at(field, fdt & 3, V + o); cc[2] = cc[2] + field_at(field, fdt & 3, 2 * V + o); }
    cc[0] = cc[0] - dz; cc[1] = cc[1] - dx; cc[2] = cc[2] - dy;
  }
  const int dims[3] = {Z, X, Y};
  bool outside = false;
#pragma unroll
  for (int a = 0; a < 3; ++a) outside = outside || !(cc[a] >= 0.0 && cc[a] <= (double)(dims[a] - 1));
  if (outside) { out[o] = out_cvt<T>(cval); return; }
  int idx[3][4]; double w[3][4];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double c = cc[a];
    const double fl = floor(c);
    const double yv = c - fl, zv = 1.0 - yv;
    w[a][1] = div6(yv * yv * (yv - 2.0) * 3.0 + 4.0);
    w[a][2] = div6(zv * zv * (zv - 2.0) * 3.0 + 4.0);
    w[a][0] = div6(zv * zv * zv);
    w[a][3] = 1.0 - w[a][0] - w[a][1] - w[a][2];
    const int st = (int)fl - 1;   // 0 <= fl <= n - 1 here
#pragma unroll
    for (int k = 0; k < 4; ++k) idx[a][k] = mirror_idx(st + k, dims[a]);
  }
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double* row = C + ((size_t)idx[0][i] * X + idx[1][j]) * Y;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double c = row[idx[2][k]];
        c = c * w[0][i]; c = c * w[1][j]; c = c * w[2][k];
        t = t + c;
      }
    }
  out[o] = out_cvt<T>(t);
}

// orders 0 and 1 on the raw stack
template <class T>
__global__ __launch_bounds__(256) void warp_lin_k(const T* __restrict__ im, int Z, int X, int Y, double dz, double dx,
                                                  double dy, const void* __restrict__ field, int fdt, int mode,
                                                  double cval, T* __restrict__ out, int order) {
  const int y = blockIdx.x * 256 + threadIdx.x;
  if (y >= Y) return;
  const int x = blockIdx.y, zq = blockIdx.z;
  const size_t o = ((size_t)zq * X + x) * Y + y, V = (size_t)Z * X * Y;
  double cc[3] = {(double)zq, (double)x, (double)y};
  if (fdt & 16) {   // (grid - drift) + field: the order of classes/preprocess.py:923-935
    cc[0] = cc[0] - dz; cc[1] = cc[1] - dx; cc[2] = cc[2] - dy;
    if (field) { cc[0] = cc[0] + field_at(field, fdt & 3, o); cc[1] = cc[1] + field_at(field, fdt & 3, V + o); cc[2] = cc[2] + field_at(field, fdt & 3, 2 * V + o); }
  } else {          // (grid + field) - drift: translate.py:19-25, io_tools/load.py:443-448
    if (field) { cc[0] = cc[0] + field_at(field, fdt & 3, o); cc[1] = cc[1] + field_at(field, fdt & 3, V + o); cc[2] = cc[2] + field_at(field, fdt & 3, 2 * V + o); }
    cc[0] = cc[0] - dz; cc[1] = cc[1] - dx; cc[2] = cc[2] - dy;
  }
  const int dims[3] = {Z, X, Y};
  if (mode == IA3_MODE_CONSTANT) {
    bool outside = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) outside = outside || cc[a] < 0.0 || cc[a] > (double)(dims[a] - 1);
    if (outside) { out[o] = out_cvt<T>(cval); return; }
  }
  if (order == 0) {   // the nearest sample: index floor(c + 0.5), clamped to the array (label images, segmentation_tools/cell.py:589)
    int id[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double r = floor(cc[a] + 0.5);
      id[a] = (int)(r < 0.0 ? 0.0 : (r > (double)(dims[a] - 1) ? (double)(dims[a] - 1) : r));
    }
    out[o] = im[((size_t)id[0] * X + id[1]) * Y + id[2]];
    return;
  }
  int st[3]; double w[3][2];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double fl = floor(cc[a]);
    const double yv = cc[a] - fl;
    // far-away coordinates must not overflow int: every index beyond the range clamps to the same edge
    st[a] = (int)(fl < -4.0 ? -4.0 : (fl > (double)dims[a] + 4.0 ? (double)dims[a] + 4.0 : fl));
    w[a][0] = 1.0 - yv; w[a][1] = yv;
  }
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        double c = (double)im[((size_t)clampi(st[0] + i, Z) * X + clampi(st[1] + j, X)) * Y + clampi(st[2] + k, Y)];
        c = c * w[0][i]; c = c * w[1][j]; c = c * w[2][k];
        t = t + c;
      }
  out[o] = out_cvt<T>(t);
}

// order 3 on the prefiltered, padded coefficient array
template <class T>
__global__ __launch_bounds__(256) void warp_cubic_k(const double* __restrict__ C, int Z, int X, int Y, double dz, double dx,
                                                    double dy, const void* __restrict__ field, int fdt,
                                                    T* __restrict__ out) {
  const int y = blockIdx.x * 256 + threadIdx.x;
  if (y >= Y) return;
  const int x = blockIdx.y, zq = blockIdx.z;
  const int Zp = Z + 2 * NPAD, Xp = X + 2 * NPAD, Yp = Y + 2 * NPAD;
  const size_t o = ((size_t)zq * X + x) * Y + y, V = (size_t)Z * X * Y;
  double cc[3] = {(double)zq, (double)x, (double)y};
  if (fdt & 16) {   // (grid - drift) + field: the order of classes/preprocess.py:923-935
    cc[0] = cc[0] - dz; cc[1] = cc[1] - dx; cc[2] = cc[2] - dy;
    if (field) { cc[0] = cc[0] + field_at(field, fdt & 3, o); cc[1] = cc[1] + field_at(field, fdt & 3, V + o); cc[2] = cc[2] + field_at(field, fdt & 3, 2 * V + o); }
  } else {          // (grid + field) - drift: translate.py:19-25, io_tools/load.py:443-448
    if (field) { cc[0] = cc[0] + field_at(field, fdt & 3, o); cc[1] = cc[1] + field_at(field, fdt & 3, V + o); cc[2] = cc[2] + field_at(field, fdt & 3, 2 * V + o); }
    cc[0] = cc[0] - dz; cc[1] = cc[1] - dx; cc[2] = cc[2] - dy;
  }
  const int dims[3] = {Zp, Xp, Yp};
  int idx[3][4]; double w[3][4];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double c = cc[a] + (double)NPAD;
    const double fl = floor(c);
    const double yv = c - fl, zv = 1.0 - yv;
    w[a][1] = div6(yv * yv * (yv - 2.0) * 3.0 + 4.0);
    w[a][2] = div6(zv * zv * (zv - 2.0) * 3.0 + 4.0);
    w[a][0] = div6(zv * zv * zv);
    w[a][3] = 1.0 - w[a][0] - w[a][1] - w[a][2];
    double f2 = fl < -8.0 ? -8.0 : (fl > (double)dims[a] + 8.0 ? (double)dims[a] + 8.0 : fl);
    const int st = (int)f2 - 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) idx[a][k] = clampi(st + k, dims[a]);
  }
  double t = 0.0;
  if ((size_t)Zp * Xp * Yp * sizeof(double) < 0xffffffffull) {
    // The padded coefficient volume is < 4 GB: buffer loads with 32-bit byte offsets from a wave-uniform resource
    // descriptor (one integer add per tap instead of 64-bit address arithmetic for each of the 64 loads, which had
    // doubled the VALU count).  When no lane of the wave clamps its y taps - always, unless a shift exceeds the 12-voxel
    // padding - the four taps of a row are 32 contiguous bytes: two 16-byte loads instead of four 8-byte ones.
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)C, (short)0, (int)((size_t)Zp * Xp * Yp * sizeof(double)), 0x00020000);
    const bool small = Xp < (1 << 24) && (unsigned)Yp * 8u < (1u << 24);
    const bool contig = idx[2][1] == idx[2][0] + 1 && idx[2][2] == idx[2][0] + 2 && idx[2][3] == idx[2][0] + 3;
    if (__all(contig)) {
      const unsigned y0 = (unsigned)idx[2][0] * 8u;
      // row offsets as sums of two products instead of 16 full 32-bit multiplies (quarter rate): z*(plane bytes) is
      // one such multiply per z tap, x*(row bytes) fits the 24-bit multiplier (x < Xp, row bytes < 2^24: checked)
      const unsigned rowb = (unsigned)Yp * 8u, planeb = (unsigned)Xp * rowb;
      unsigned zo[4], xo[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { zo[i] = (unsigned)idx[0][i] * planeb + y0; xo[i] = small ? __umul24((unsigned)idx[1][i], rowb) : (unsigned)idx[1][i] * rowb; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned off = zo[i] + xo[j];
          const v4u lo = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
          const v4u hi = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16u, 0, 0);
          const double cv[4] = {__hiloint2double((int)lo.y, (int)lo.x), __hiloint2double((int)lo.w, (int)lo.z),
                                __hiloint2double((int)hi.y, (int)hi.x), __hiloint2double((int)hi.w, (int)hi.z)};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            double c = cv[k];
            c = c * w[0][i]; c = c * w[1][j]; c = c * w[2][k];
            t = t + c;
          }
        }
    } else {
      unsigned yoff[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) yoff[k] = (unsigned)idx[2][k] * 8u;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned roff = ((unsigned)idx[0][i] * (unsigned)Xp + (unsigned)idx[1][j]) * (unsigned)Yp * 8u;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const v2u q = __builtin_amdgcn_raw_buffer_load_b64(rsrc, roff + yoff[k], 0, 0);
            double c = __hiloint2double((int)q.y, (int)q.x);
            c = c * w[0][i]; c = c * w[1][j]; c = c * w[2][k];
            t = t + c;
          }
        }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double* row = C + ((size_t)idx[0][i] * Xp + idx[1][j]) * Yp;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          double c = row[idx[2][k]];
          c = c * w[0][i]; c = c * w[1][j]; c = c * w[2][k];
          t = t + c;
        }
      }
  }
  out[o] = out_cvt<T>(t);
}

// order 3, NV (2 or 4) consecutive outputs of a row per thread.  warp_cubic_k reads 64 coefficients (512 bytes) per
// output.  Neighbouring outputs of a row share three of their four taps in every one of the 16 (z, x) rows whenever their
// coordinates fall into consecutive cells — the normal case for a drift plus a smooth field — so a thread that makes
// NV outputs loads a run of NV + 3 coefficients per row instead of NV x 4: 320 (NV = 2) or 224 (NV = 4) bytes per
// output.  Every output still sums ITS 64 products in SciPy's order with its own weights.  An output whose cell does not line up with the first one's (a coordinate crosses an
// integer inside the group), or a run that would leave the padded row, takes the per-tap loads of warp_cubic_k.
template <class T, int NV, int OCC>
__global__ __launch_bounds__(256, OCC) void warp_cubic4_k(const double* __restrict__ C, int Z, int X, int Y, double dz, double dx,
                                                     double dy, const void* __restrict__ field, int fdt,
                                                     T* __restrict__ out, int rows_per) {
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  // Block order: an output row needs 4 x 4 coefficient rows, and its neighbours along z and x need mostly the same ones.
  // Blocks go to the eight XCDs in turn and every XCD has an L2 of its own, so XCD c takes the rows of slab c
  // (X / 8 consecutive rows), walks them with z fastest, then x, then the piece of the row: what the blocks in flight
  // on an XCD need (two or three rows x all planes, ~3 MB) stays in its L2, and a coefficient comes from HBM about once
  // (the plain (y, x, z) order fetched 2.8 x the coefficient stack).
  const unsigned b = blockIdx.x, xcd = b & 7u, bi = b >> 3;
  const unsigned zq_u = bi % (unsigned)Z, r1 = bi / (unsigned)Z;
  const unsigned xl = r1 % (unsigned)rows_per, h = r1 / (unsigned)rows_per;
  const int x = (int)(xcd * (unsigned)rows_per + xl), zq = (int)zq_u;
  static_assert(NV == 2 || NV == 4, "outputs per thread");
  constexpr int RUN = NV + 3;   // coefficients of a row that NV consecutive outputs share
  const int y0 = (int)(h * 256u + threadIdx.x) * NV;
  if (x >= X || y0 >= Y) return;       // Y % NV == 0 (host)
  const int Zp = Z + 2 * NPAD, Xp = X + 2 * NPAD, Yp = Y + 2 * NPAD;
  const size_t o = ((size_t)zq * X + x) * Y + y0, V = (size_t)Z * X * Y;
  double f[3][NV];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int v = 0; v < NV; ++v) f[a][v] = 0.0;
  if (field) {
    typedef float fvec __attribute__((ext_vector_type(NV)));
    typedef double dvec __attribute__((ext_vector_type(2)));
    if ((fdt & 3) == 1) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const fvec q = *(const fvec*)((const float*)field + (size_t)a * V + o);
#pragma unroll
        for (int v = 0; v < NV; ++v) f[a][v] = (double)q[v];
      }
    } else {
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int v = 0; v < NV; v += 2) {
          const dvec q = *(const dvec*)((const double*)field + (size_t)a * V + o + v);
          f[a][v] = q.x; f[a][v + 1] = q.y;
        }
    }
  }
  const int dims[3] = {Zp, Xp, Yp};
  const double dr[3] = {dz, dx, dy};
  double w[NV][3][4];    // [output][axis][tap]
  int st[NV][3];         // first tap (before clamping)
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    double cc[3] = {(double)zq, (double)x, (double)(y0 + v)};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (fdt & 16) { cc[a] = cc[a] - dr[a]; if (field) cc[a] = cc[a] + f[a][v]; }   // (grid - drift) + field
      else { if (field) cc[a] = cc[a] + f[a][v]; cc[a] = cc[a] - dr[a]; }            // (grid + field) - drift
      const double c = cc[a] + (double)NPAD;
      const double fl = floor(c);
      const double yv = c - fl, zv = 1.0 - yv;
      w[v][a][1] = div6(yv * yv * (yv - 2.0) * 3.0 + 4.0);
      w[v][a][2] = div6(zv * zv * (zv - 2.0) * 3.0 + 4.0);
      w[v][a][0] = div6(zv * zv * zv);
      w[v][a][3] = 1.0 - w[v][a][0] - w[v][a][1] - w[v][a][2];
      const double f2 = fl < -8.0 ? -8.0 : (fl > (double)dims[a] + 8.0 ? (double)dims[a] + 8.0 : fl);
      st[v][a] = (int)f2 - 1;
    }
  }
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)C, (short)0, (int)((size_t)Zp * Xp * Yp * sizeof(double)), 0x00020000);
  const unsigned rowb = (unsigned)Yp * 8u, planeb = (unsigned)Xp * rowb;
  double t[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) t[v] = 0.0;
  const bool run_ok = st[0][2] >= 0 && st[0][2] + RUN - 1 <= Yp - 1;
  if (run_ok) {
    unsigned zo[4], xo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      zo[i] = (unsigned)clampi(st[0][0] + i, Zp) * planeb + (unsigned)st[0][2] * 8u;
      xo[i] = (unsigned)clampi(st[0][1] + i, Xp) * rowb;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned off = zo[i] + xo[j];
        double r[RUN + 1];
#pragma unroll
        for (int m = 0; m + 1 < RUN; m += 2) {
          const v4u q = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 8u * m, 0, 0);
          r[m] = __hiloint2double((int)q.y, (int)q.x); r[m + 1] = __hiloint2double((int)q.w, (int)q.z);
        }
        {
          const v2u q = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off + 8u * (RUN - 1), 0, 0);   // RUN is odd
          r[RUN - 1] = __hiloint2double((int)q.y, (int)q.x);
        }
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            double c = r[v + k];
            c = c * w[v][0][i]; c = c * w[v][1][j]; c = c * w[v][2][k];
            t[v] = t[v] + c;
          }
      }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const bool lined_up = run_ok && st[v][0] == st[0][0] && st[v][1] == st[0][1] && st[v][2] == st[0][2] + v;
    if (!lined_up) {
      unsigned yoff[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) yoff[k] = (unsigned)clampi(st[v][2] + k, Yp) * 8u;
      double tv = 0.0;
#pragma unroll 1
      for (int i = 0; i < 4; ++i) {
        const unsigned zoff = (unsigned)clampi(st[v][0] + i, Zp) * planeb;
        const double wz = i == 0 ? w[v][0][0] : (i == 1 ? w[v][0][1] : (i == 2 ? w[v][0][2] : w[v][0][3]));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned roff = zoff + (unsigned)clampi(st[v][1] + j, Xp) * rowb;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const v2u q = __builtin_amdgcn_raw_buffer_load_b64(rsrc, roff + yoff[k], 0, 0);
            double c = __hiloint2double((int)q.y, (int)q.x);
            c = c * wz; c = c * w[v][1][j]; c = c * w[v][2][k];
            tv = tv + c;
          }
        }
      }
      t[v] = tv;
    }
  }
  typedef T tvec __attribute__((ext_vector_type(NV)));
  tvec res;
#pragma unroll
  for (int v = 0; v < NV; ++v) res[v] = out_cvt<T>(t[v]);
  *(tvec*)(out + o) = res;
}

int g_warp_warm = 64;   // IA3_TUNE_WARP_ONEPASS

// pass: 0, 1, 2 = how many passes of the prefilter the samples have been through; amax_bits: see IirInit
IirInit make_init(int n, int pass, double src_max, const unsigned* amax_bits) {
  IirInit q;
  q.z = IA3_POLE3;
  q.gain = (1.0 - q.z) * (1.0 - 1.0 / q.z);
  q.zn = pow(q.z, (double)n);
  q.scale = q.z / (1.0 - q.zn * q.zn);
  q.full = q.zn != 0.0 ? 1 : 0;
  q.bound = 1.001 * q.gain * pow(3.0, (double)pass) * (amax_bits ? 1.0 : src_max);
  q.amax_bits = amax_bits;
  return q;
}

template <class T>
int warp_t(const ia3_stack* im, const double* drift, const void* field, int fdt, int order, int mode, double cval,
           ia3_stack* out) {
  hipStream_t st = stream();
  const int Z = im->Z, X = im->X, Y = im->Y;
  dim3 g((unsigned)((Y + 255) / 256), (unsigned)X, (unsigned)Z);
  if (order <= 1) {
    ProfScope ps("warp_linear");
    hipLaunchKernelGGL((warp_lin_k<T>), g, dim3(256), 0, st, (const T*)im->d, Z, X, Y, drift[0], drift[1], drift[2],
                       field, fdt, mode, cval, (T*)out->d, order);
    IA3_KCHECK();
    return IA3_OK;
  }
  if (mode == IA3_MODE_CONSTANT) {
    const size_t V = (size_t)Z * X * Y, plane = (size_t)X * Y;
    Scratch P(V * sizeof(double));
    if (!P.p) return IA3_ENOMEM;
    auto init = [](int n) {
      MirInit q;
      q.z = IA3_POLE3;
      q.gain = (1.0 - q.z) * (1.0 - 1.0 / q.z);
      q.zn1 = pow(q.z, (double)(n - 1));
      q.den = 1.0 - q.zn1 * q.zn1;
      double zi = q.z;
      int i = 1;
      while (zi != 0.0 && i < n) { zi *= q.z; ++i; }
      q.n_sum = i;
      return q;
    };
    {
      ProfScope ps("spline_mirror");
      hipLaunchKernelGGL((spline_cvt_k<T>), dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, (const T*)im->d, V, P.as<double>());
      hipLaunchKernelGGL(spline_mirror_k, dim3((unsigned)((plane + 255) / 256), 1), dim3(256), 0, st, P.as<double>(), (int)plane,
                         (size_t)1, plane, Z, (size_t)0, init(Z));
      hipLaunchKernelGGL(spline_mirror_k, dim3((unsigned)((Y + 255) / 256), (unsigned)Z), dim3(256), 0, st, P.as<double>(), Y,
                         (size_t)1, (size_t)Y, X, plane, init(X));
      hipLaunchKernelGGL(spline_mirror_k, dim3((unsigned)((X + 255) / 256), (unsigned)Z), dim3(256), 0, st, P.as<double>(), X,
                         (size_t)Y, (size_t)1, Y, plane, init(Y));
    }
    ProfScope ps("warp_cubic");
    hipLaunchKernelGGL((warp_cubic_mirror_k<T>), g, dim3(256), 0, st, (const double*)P.as<double>(), Z, X, Y, drift[0], drift[1],
                       drift[2], field, fdt, cval, (T*)out->d);
    IA3_KCHECK();
    return IA3_OK;
  }
  const int Zp = Z + 2 * NPAD, Xp = X + 2 * NPAD, Yp = Y + 2 * NPAD;
  const size_t coef_bytes = (size_t)Zp * Xp * Yp * sizeof(double);
  Scratch P(coef_bytes + 256);
  if (!P.p) return IA3_ENOMEM;
  const size_t plane = (size_t)Xp * Yp;
  const unsigned* amax = nullptr;
  Scratch zrs((3 * ZR_BLOCKS + 8) * sizeof(float));   // partials of the field scan | zr[2]
  if (!zrs.p) return IA3_ENOMEM;
  int* zr = (int*)((float*)zrs.p + 3 * ZR_BLOCKS);
  {
    ProfScope ps("spline_axis0");
    {   // the planes this warp's gather can read (zrange_k)
      const size_t V = (size_t)Z * X * Y;
      const float* src3 = (const float*)zrs.p;   // [lo, hi, bad] triples
      int nb = 1;
      auto scan = [&]() {
        nb = (int)((V / 4 + 255) / 256 < (size_t)ZR_BLOCKS ? (V / 4 + 255) / 256 : (size_t)ZR_BLOCKS);
        if (nb < 1) nb = 1;
        if ((fdt & 3) == 1) hipLaunchKernelGGL((field_zminmax_k<float>), dim3((unsigned)nb), dim3(256), 0, st, (const float*)field, V, (float*)zrs.p);
        else hipLaunchKernelGGL((field_zminmax_k<double>), dim3((unsigned)nb), dim3(256), 0, st, (const double*)field, V, (float*)zrs.p);
      };
      if (!field) {
        IA3_HIP(hipMemsetAsync(zrs.p, 0, 3 * sizeof(float), st));   // one triple: lo = hi = 0, nothing bad
      } else {
        // a field that lives in a buffer of ia3_buffer_upload is constant over the run: its range is computed once and kept
        // beside it (the scan reads 0.84 GB for a 50 x 2048 x 2048 float32 field: 0.2 ms of every warp otherwise)
        bool ready = false, fill = false;
        float* note = const_note(field, &ready, &fill);
        if (note && ready) src3 = note;
        else {
          scan();
          float* folded = note && fill ? note : (float*)zrs.p + 3 * ZR_BLOCKS + 4;   // (behind the two words of zr)
          hipLaunchKernelGGL(zminmax_fold_k, dim3(1), dim3(256), 0, st, (const float*)zrs.p, nb, folded);
          src3 = folded;
          nb = 1;
          if (note && fill) {
            const bool ok = hipStreamSynchronize(st) == hipSuccess;   // once per field: other streams may read the note from now on
            const_note_filled(field, ok);
          }
        }
      }
      hipLaunchKernelGGL(zrange_k, dim3(1), dim3(1), 0, st, src3, nb, drift[0], Z, zr);
    }
    // largest |sample| of the source, for the cut of the start sums along long axes (IirInit)
    if (sizeof(T) == 4) {
      unsigned* slot = (unsigned*)((char*)P.p + coef_bytes);
      IA3_HIP(hipMemsetAsync(slot, 0, sizeof(unsigned), st));
      hipLaunchKernelGGL(absmax_f32_k, dim3(2048), dim3(256), 0, st, (const float*)im->d, (size_t)Z * X * Y, slot);
      amax = slot;
    }
    // axis 0: lines = (x,y) columns, stride = plane
    IirInit qz = make_init(Zp, 0, 65535.0, amax);
    qz.full = 1;  // the faithful start sum is always affordable along the short z axis
    bool done0 = false;
    switch (Z) {
#define IA3_WARP_DEPTH(D)                                                                                              \
      case D:                                                                                                          \
        hipLaunchKernelGGL((spline_pad_iir0_n_k<T, D + 2 * NPAD>), dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, st, \
                           (const T*)im->d, X, Y, P.as<double>(), qz, (const int*)zr);                                 \
        done0 = true;                                                                                                  \
        break;
      IA3_WARP_DEPTHS(IA3_WARP_DEPTH)
#undef IA3_WARP_DEPTH
      default: break;
    }
    if (!done0 && Z >= 8 && Z <= 64) {
      // a depth without an instantiation of its own: the same kernel from the run-time compiler (rtc.cpp; under a second, once
      // per depth, dtype and machine, then from the cache file); not available -> the generic kernels below
      char name[96];
      snprintf(name, sizeof(name), "ia3warpk::spline_pad_iir0_n_k<%s, %d>", sizeof(T) == 4 ? "float" : "unsigned short", Z + 2 * NPAD);
      std::vector<hipFunction_t> fns;
      if (rtc_kernels("warp0", {"warp_iir0_kernel.inc"}, "#include <stdint.h>\n", {name}, fns)) {
        const T* a_im = (const T*)im->d;
        int a_x = X, a_y = Y;
        double* a_p = P.as<double>();
        IirInit a_q = qz;
        const int* a_zr = zr;
        void* args[] = {(void*)&a_im, (void*)&a_x, (void*)&a_y, (void*)&a_p, (void*)&a_q, (void*)&a_zr};
        if (hipModuleLaunchKernel(fns[0], (unsigned)((plane + 255) / 256), 1, 1, 256, 1, 1, 0, st, args, nullptr) != hipSuccess)
          return set_error(IA3_EHIP, "launch of the run-time compiled prefilter kernel (depth %d) failed: %s", Z, hipGetErrorString(hipGetLastError()));
        done0 = true;
      }
    }
    if (done0) {
    } else if (Zp <= 80) {   // padded line fits in registers: pad + axis-0 recursion in one pass
      hipLaunchKernelGGL((spline_pad_iir0_k<T, 80>), dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, st, (const T*)im->d, Z, X, Y,
                         P.as<double>(), qz, (const int*)zr);
    } else {
      hipLaunchKernelGGL((spline_pad_k<T>), dim3((unsigned)((Yp + 255) / 256), (unsigned)Xp, (unsigned)Zp), dim3(256), 0, st,
                         (const T*)im->d, Z, X, Y, P.as<double>());
      hipLaunchKernelGGL(spline_iir_strided_k, dim3((unsigned)((plane + 255) / 256), 1), dim3(256), 0, st, P.as<double>(),
                         (int)plane, plane, Zp, (size_t)0, qz, (const int*)nullptr);
    }
  }
  {   // axis 1: lines = (z,y), stride = Yp
    ProfScope ps("spline_axis1");
    IirInit qx = make_init(Xp, 1, 65535.0, amax);
    // T = 12 samples per step, K = 52 of warm-up: 64 doubles of window, three waves per SIMD without spills (measured on
    // 74 x 2072 x 2072: <16,48> 1.22 ms but its chains meet too late for dim uint16 backgrounds, <12,52> 1.34, <8,56> 1.35,
    // <16,64> at two waves per SIMD 1.43, <8,64> 1.70 and <32,64> 1.70 with spills; two sweeps 2.28)
    constexpr int T1 = 12, K1 = 52;
    if (g_warp_warm > 0 && Xp >= 2 * (T1 + K1))
      hipLaunchKernelGGL((spline_iir_strided_1p_k<T1, K1, 3>), dim3((unsigned)((Yp + 63) / 64), (unsigned)Zp), dim3(64), 0, st,
                         P.as<double>(), Yp, (size_t)Yp, Xp, plane, qx, g_warp_warm < K1 ? g_warp_warm : K1, (const int*)zr);
    else
      hipLaunchKernelGGL(spline_iir_strided_k, dim3((unsigned)((Yp + 255) / 256), (unsigned)Zp), dim3(256), 0, st, P.as<double>(),
                         Yp, (size_t)Yp, Xp, plane, qx, (const int*)zr);
  }
  {   // axis 2: contiguous lines (z,x)
    ProfScope ps("spline_axis2");
    IirInit qy = make_init(Yp, 2, 65535.0, amax);
    const size_t nl = (size_t)Zp * Xp;
    hipLaunchKernelGGL((spline_iir_contig_k<64, 8, 3>), dim3((unsigned)((nl + 31) / 32)), dim3(256), 0, st, P.as<double>(), nl,
                       Yp, qy, g_warp_warm, (const int*)zr, Xp);
  }
  {
    ProfScope ps("warp_cubic");
    const bool aligned = Y % 2 == 0 && ((uintptr_t)out->d & 7) == 0 && ((uintptr_t)field & 15) == 0;
    if (aligned && coef_bytes < 0xffffffffull && g_warp_warm >= 0)
    {
      // two outputs per thread at five waves per SIMD (94 registers): 3.66 ms on 50 x 2048 x 2048; four per thread need 168
      // registers (three waves): 4.0 ms — 5.7 ms when forced to 128 (spills), 5.0 ms at two waves — although they issue
      // fewer loads: the kernel lives on waves in flight, not on its instruction count (a certified fused multiply-add for
      // uint16 outputs — 192 instead of 256 operations per output, outputs within 1e-6 of a rounding boundary recomputed —
      // took 10 % of the instructions off and nothing off the time, and was taken out again)
      const int rows_per = (X + 7) / 8, nh = (Y / 2 + 255) / 256;
      hipLaunchKernelGGL((warp_cubic4_k<T, 2, 5>), dim3((unsigned)(8 * rows_per * Z * nh)), dim3(256), 0, st,
                         (const double*)P.as<double>(), Z, X, Y, drift[0], drift[1], drift[2], field, fdt, (T*)out->d, rows_per);
    }
    else
      hipLaunchKernelGGL((warp_cubic_k<T>), g, dim3(256), 0, st, (const double*)P.as<double>(), Z, X, Y, drift[0], drift[1],
                         drift[2], field, fdt, (T*)out->d);
  }
  IA3_KCHECK();
  return IA3_OK;
}

}  // namespace

namespace ia3k { void set_warp_onepass(int v) { g_warp_warm = v; } }

extern "C" {

// field: NULL or device pointer to a (3,Z,X,Y) displacement field; field_dtype 1 = float32, 2 = float64
int ia3_warp3d_dev(const ia3_stack* im, const double* drift, const void* field_dev, int field_dtype, int order,
                   int mode, double cval, ia3_stack* out) {
  int rc = ensure_init(); if (rc) return rc;
  if (!im || !out || !drift) return set_error(IA3_EINVAL, "null argument");
  if (im->dtype != out->dtype || im->Z != out->Z || im->X != out->X || im->Y != out->Y || im->d == out->d)
    return set_error(IA3_EINVAL, "output stack must be a distinct stack of the same shape and dtype");
  if (order != 0 && order != 1 && order != 3) return set_error(IA3_EUNSUPPORTED, "warp order %d (0, 1 and 3 are implemented)", order);
  if (order <= 1 && mode != IA3_MODE_CONSTANT && mode != IA3_MODE_NEAREST) return set_error(IA3_EUNSUPPORTED, "border mode %d", mode);
  if (order == 3 && mode != IA3_MODE_NEAREST && mode != IA3_MODE_CONSTANT) return set_error(IA3_EUNSUPPORTED, "border mode %d", mode);
  if (order == 3 && mode == IA3_MODE_CONSTANT && (im->Z < 2 || im->X < 2 || im->Y < 2))
    return set_error(IA3_EUNSUPPORTED, "order 3 with mode 'constant' needs at least two samples along every axis");
  if (field_dev && (field_dtype & ~16) != 1 && (field_dtype & ~16) != 2) return set_error(IA3_EINVAL, "field dtype must be float32 (1) or float64 (2), optionally + 16");
  if (im->dtype == IA3_F32) return warp_t<float>(im, drift, field_dev, field_dtype, order, mode, cval, out);
  return warp_t<uint16_t>(im, drift, field_dev, field_dtype, order, mode, cval, out);
}

int ia3_warp3d(const void* im, int dtype, int Z, int X, int Y, const double* drift, const void* field, int field_dtype,
               int order, int mode, double cval, void* out) {
  ia3_stack *a = nullptr, *b = nullptr;
  int rc = ia3_stack_upload(im, dtype, Z, X, Y, &a); if (rc) return rc;
  rc = ia3_stack_alloc(dtype, Z, X, Y, &b);
  void* dfield = nullptr;
  if (!rc && field) {
    size_t bytes = (size_t)3 * Z * X * Y * ((field_dtype & 3) == 1 ? 4 : 8);
    if (hipMalloc(&dfield, bytes) != hipSuccess) rc = set_error(IA3_ENOMEM, "hipMalloc(%zu) for the displacement field failed", bytes);
    else if (hipMemcpy(dfield, field, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = set_error(IA3_EHIP, "field upload failed");
  }
  if (!rc) rc = ia3_warp3d_dev(a, drift, dfield, field_dtype, order, mode, cval, b);
  if (!rc) rc = ia3_stack_download(b, out);
  if (dfield) (void)hipFree(dfield);
  ia3_stack_free(a); ia3_stack_free(b);
  return rc;
}

}  // extern "C"
