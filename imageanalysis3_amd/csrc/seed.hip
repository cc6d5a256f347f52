// DoG local-maximum seeding (reference: spot_tools/fitting.py:20-165 get_seeds) for gfx950.
//
//   max_im = G_front(im), min_im = G_back(im)            (gauss.hip, exact ndimage arithmetic)
//   mask   = (maxfilt_s(max_im) == max_im) & (minfilt_s(min_im) != min_im)      fitting.py:95,102
//   diff   = float32(max_im) - float32(min_im)                                  fitting.py:106
//   keep   diff >= th_i, d <= c <= size-d on every axis                         fitting.py:113-125,156-165
// The candidate list is made at the LOWEST dynamic threshold level, so the dynamic-threshold loop (pick the first level
// with enough seeds) needs no second sweep.  Candidates are sparse (~1e-5 of the voxels): a wave ballots its hits, one lane
// reserves space with a single atomic and the hits are written by prefix rank (wavefront ballot + prefix-sum
// compaction).  The rest (np.where order, hot-column vote, sort by height, truncation: fitting.py:131-150) runs on the
// device too (fin_*_k) for up to 32768 candidates, on the host beyond.
//
// Two forms of the detector.  Default (3x3x3 footprint, background radius 4..63): the background filter is evaluated
// LAZILY — only its axis-0 pass on the whole stack, a block-minimum bound of min_im, candidates = local maxima of max_im
// that can still reach the lowest level, exact axis-1 / axis-2 passes at those (see "lazy background filter" below).
// Otherwise, and as the fallback when the lazy form's candidate list overflows: both filtered stacks are read once
// (8 B/voxel for f32) by an LDS-tiled kernel with a rolling three-plane register pipeline along z.
#include "ia3_rt.h"
#include <memory>
#include <algorithm>
#include <math.h>
#include <string.h>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>

namespace {

struct Cand { int z, x, y; float h; };   // 16 B

constexpr int MAXLEV = 64;
struct Levels { double th[MAXLEV]; int n; };

struct SeedCtl {            // device-resident counters (16 B so the candidate array stays 16-B aligned)
  unsigned int n_cand;      // candidates found (may exceed capacity)
  unsigned int overflow;
  unsigned int pad[2];
};

template <class T> __device__ __forceinline__ T ldc(const T* p, int z, int x, int y, int X, int Y) {
  return p[((size_t)z * X + x) * Y + y];
}

// in-plane window max/min over [x+LO, x+HI] x [y+LO, y+HI] clipped to the image.  Clipping equals
// scipy's default 'reflect' border for a rank filter: every reflected index is already in the window.
template <class T, bool IS_MAX, int LO, int HI>
__device__ __forceinline__ T plane_ext(const T* p, int z, int x, int y, int X, int Y) {
  T r = ldc(p, z, x, y, X, Y);
#pragma unroll
  for (int dx = LO; dx <= HI; ++dx) {
    int xx = min(max(x + dx, 0), X - 1);
#pragma unroll
    for (int dy = LO; dy <= HI; ++dy) {
      int yy = min(max(y + dy, 0), Y - 1);
      T v = ldc(p, z, xx, yy, X, Y);
      r = IS_MAX ? (v > r ? v : r) : (v < r ? v : r);
    }
  }
  return r;
}

// ZC output planes per thread; window of W taps [LO, HI] along every axis (W=3 -> -1..1).
template <class T, int ZC, int W>
__global__ __launch_bounds__(256) void seed_detect(const T* __restrict__ mx, const T* __restrict__ mn,
                                                   int Z, int X, int Y, int edge, double th_low,
                                                   Cand* __restrict__ out, unsigned capacity,
                                                   SeedCtl* __restrict__ ctl, int rule) {
  constexpr int LO = -(W / 2), HI = W - W / 2 - 1;
  const int y = blockIdx.x * 64 + (threadIdx.x & 63);
  const int x = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int z0 = blockIdx.z * ZC;
  const bool inside = (x < X) && (y < Y);
  const int xs = inside ? x : 0, ys = inside ? y : 0;
  T pmax[W], pmin[W];  // plane extrema for z+LO .. z+HI (rolling, statically indexed)
#pragma unroll
  for (int k = 0; k < W - 1; ++k) {
    int zz = min(max(z0 + LO + k, 0), Z - 1);
    pmax[k] = plane_ext<T, true, LO, HI>(mx, zz, xs, ys, X, Y);
    pmin[k] = plane_ext<T, false, LO, HI>(mn, zz, xs, ys, X, Y);
  }
  for (int z = z0; z < z0 + ZC && z < Z; ++z) {
    int zz = min(z + HI, Z - 1);
    pmax[W - 1] = plane_ext<T, true, LO, HI>(mx, zz, xs, ys, X, Y);
    pmin[W - 1] = plane_ext<T, false, LO, HI>(mn, zz, xs, ys, X, Y);
    T vmax = pmax[0], vmin = pmin[0];
#pragma unroll
    for (int k = 1; k < W; ++k) { vmax = pmax[k] > vmax ? pmax[k] : vmax; vmin = pmin[k] < vmin ? pmin[k] : vmin; }
#pragma unroll
    for (int k = 0; k < W - 1; ++k) { pmax[k] = pmax[k + 1]; pmin[k] = pmin[k + 1]; }
    const T cmax = ldc(mx, z, xs, ys, X, Y), cmin = ldc(mn, z, xs, ys, X, Y);
    float diff = (float)cmax - (float)cmin;
    bool hit = inside && (vmax == cmax) && (vmin != cmin) && ((double)diff >= th_low);
    if (rule == 1) {
      // legacy visual_tools.py:362-369 get_seed_points_base: the rank-filter outputs are cast to int64
      // (truncation) before the equality tests, the background minimum must be non-zero, and the height
      // is the integer difference, compared with a strict '>'
      const long long imax = (long long)vmax, imin = (long long)vmin;
      const long long hh = imax - imin;
      hit = inside && ((double)imax == (double)cmax) && ((double)imin != (double)cmin) && (imin != 0) &&
            ((double)hh > th_low);
      diff = (float)hh;
    }
    if (edge > 0)
      hit = hit && z >= edge && z <= Z - edge && x >= edge && x <= X - edge && y >= edge && y <= Y - edge;
    const unsigned long long ballot = __ballot(hit);
    if (ballot) {  // wave-uniform: ballot + prefix-rank compaction, one atomic per wave
      const int lane = threadIdx.x & 63;
      unsigned basepos = 0;
      if (lane == 0) basepos = atomicAdd(&ctl->n_cand, (unsigned)__popcll(ballot));
      basepos = __shfl(basepos, 0);
      if (hit) {
        unsigned pos = basepos + (unsigned)__popcll(ballot & ((1ull << lane) - 1ull));
        if (pos < capacity) out[pos] = Cand{z, x, y, diff};
        else ctl->overflow = 1;
      }
    }
  }
}

// ---- fast path for the 3x3x3 window: LDS-tiled planes, rolling along z ---------------------------------------
// A 256-thread block owns a 16 (x) x 64 (y) column of the stack and marches along z.  Per plane it stages the
// (16+2) x (64+2) halo tile of both filtered stacks in LDS with coalesced row loads, every thread takes the 3x3
// in-plane extrema of its 4 voxels from LDS, and the z direction is a three-deep register pipeline.  Each voxel is
// fetched from HBM 1.16x (halo) instead of 2.3x.
// Workgroups are handed to the 8 XCDs round-robin by linear id, and each XCD has its own L2: tiles that share halo
// lines must land on the same XCD to share them.  Map linear id b to the tile (b % 8) * ceil(n/8) + b / 8, so every
// XCD walks a contiguous run of tiles (neighbours in y are 8 dispatch slots apart, resident together).  Returns -1
// for the padding ids of the last run.
__device__ __forceinline__ int xcd_tile(int b, int n_tiles) {
  const int per = (n_tiles + 7) >> 3;
  const int t = (b & 7) * per + (b >> 3);
  return ((b >> 3) < per && t < n_tiles) ? t : -1;
}

// three-input extrema: one v_max3 / v_min3 instruction each (the `a > b ? a : b` spelling compiles to a compare, a
// select and a hazard nop per pair: the SQ counters showed 50 VALU instructions per voxel, most of them these)
__device__ __forceinline__ float max3v(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float min3v(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
__device__ __forceinline__ uint16_t max3v(uint16_t a, uint16_t b, uint16_t c) {
  const unsigned m = max(max((unsigned)a, (unsigned)b), (unsigned)c);
  return (uint16_t)m;
}
__device__ __forceinline__ uint16_t min3v(uint16_t a, uint16_t b, uint16_t c) {
  const unsigned m = min(min((unsigned)a, (unsigned)b), (unsigned)c);
  return (uint16_t)m;
}

template <class T, int ZC>
__global__ __launch_bounds__(256) void seed_detect3_tiled(const T* __restrict__ mx, const T* __restrict__ mn,
                                                          int Z, int X, int Y, int edge, double th_low,
                                                          Cand* __restrict__ out, unsigned capacity,
                                                          SeedCtl* __restrict__ ctl) {
  constexpr int TX = 16, TY = 64, HX = TX + 2, HY = TY + 2, NE = (HX * HY + 255) / 256;
  __shared__ T tmax[2][HX][HY + 2];   // double-buffered: plane q+1 is fetched while plane q is consumed
  __shared__ T tmin[2][HX][HY + 2];
  const int nty = (Y + TY - 1) / TY, ntx = (X + TX - 1) / TX;
  const int tile = xcd_tile(blockIdx.x, nty * ntx);   // grid.x = 8 * ceil(tiles / 8)
  if (tile < 0) return;                               // whole block leaves together
  const int x0 = (tile / nty) * TX, y0 = (tile % nty) * TY;
  const int z0 = blockIdx.z * ZC, z1 = z0 + ZC < Z ? z0 + ZC : Z;
  const int ty = threadIdx.x & 63, tg = threadIdx.x >> 6;   // thread owns rows tg*4 .. tg*4+3, column ty
  // per-thread staging slots: element e = threadIdx.x + 256*i of the halo tile -> in-plane offset / LDS slot
  size_t goff[NE]; int lr[NE], lc[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = threadIdx.x + 256 * i;
    const int r = e / HY, c = e % HY;
    lr[i] = e < HX * HY ? r : -1; lc[i] = c;
    const int gx = min(max(x0 + r - 1, 0), X - 1), gy = min(max(y0 + c - 1, 0), Y - 1);
    goff[i] = (size_t)gx * Y + gy;
  }
  T ra[NE], rb[NE];
  auto fetch = [&](int q) {
    const int zq = q < 0 ? 0 : (q >= Z ? Z - 1 : q);
    const size_t pz = (size_t)zq * X * Y;
#pragma unroll
    for (int i = 0; i < NE; ++i) if (lr[i] >= 0) { ra[i] = mx[pz + goff[i]]; rb[i] = mn[pz + goff[i]]; }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NE; ++i) if (lr[i] >= 0) { tmax[buf][lr[i]][lc[i]] = ra[i]; tmin[buf][lr[i]][lc[i]] = rb[i]; }
  };
  T pM[4][3], pm[4][3], cM[4][2], cm[4][2];                 // plane extrema (z-1,z,z+1) and centres (z, z+1)
#pragma unroll
  for (int v = 0; v < 4; ++v) { pM[v][0] = pM[v][1] = pM[v][2] = 0; pm[v][0] = pm[v][1] = pm[v][2] = 0; cM[v][0] = cM[v][1] = 0; cm[v][0] = cm[v][1] = 0; }
  fetch(z0 - 1);
  stash(0);
  __syncthreads();
  // planes are visited from z0-1 to z1 (clamped); after visiting plane q the pipeline holds q-2, q-1, q
  for (int q = z0 - 1, buf = 0; q <= z1; ++q, buf ^= 1) {
    if (q < z1) fetch(q + 1);   // in flight while plane q is reduced from LDS
    {
      // separable 3x3: row extrema of the 6 halo rows this thread's 4 voxels touch (3 LDS reads each), then 3-row
      // extrema per voxel — half the LDS reads of the direct 9-tap form
      const int c = ty + 1;
      T hM[6], hm[6], ce[4], cf[4];
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) {
        const int r = tg * 4 + rr;
        const T a0 = tmax[buf][r][c - 1], a1 = tmax[buf][r][c], a2 = tmax[buf][r][c + 1];
        const T b0 = tmin[buf][r][c - 1], b1 = tmin[buf][r][c], b2 = tmin[buf][r][c + 1];
        const T a = max3v(a0, a1, a2), b = min3v(b0, b1, b2);
        hM[rr] = a; hm[rr] = b;
        if (rr >= 1 && rr <= 4) { ce[rr - 1] = a1; cf[rr - 1] = b1; }
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const T a = max3v(hM[v], hM[v + 1], hM[v + 2]), b = min3v(hm[v], hm[v + 1], hm[v + 2]);
        pM[v][0] = pM[v][1]; pM[v][1] = pM[v][2]; pM[v][2] = a;
        pm[v][0] = pm[v][1]; pm[v][1] = pm[v][2]; pm[v][2] = b;
        cM[v][0] = cM[v][1]; cM[v][1] = ce[v];
        cm[v][0] = cm[v][1]; cm[v][1] = cf[v];
      }
    }
    if (q < z1) stash(buf ^ 1);   // the other buffer was last read one iteration ago, before the previous barrier
    __syncthreads();
    const int z = q - 1;   // plane whose 3-plane window is now complete
    if (z < z0) continue;  // (uniform) pipeline still filling
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int x = x0 + tg * 4 + v, y = y0 + ty;
      const T vmax = max3v(pM[v][0], pM[v][1], pM[v][2]), vmin = min3v(pm[v][0], pm[v][1], pm[v][2]);
      const T cmax = cM[v][0], cmin = cm[v][0];
      const float diff = (float)cmax - (float)cmin;
      bool hit = x < X && y < Y && (vmax == cmax) && (vmin != cmin) && ((double)diff >= th_low);
      if (edge > 0)
        hit = hit && z >= edge && z <= Z - edge && x >= edge && x <= X - edge && y >= edge && y <= Y - edge;
      const unsigned long long ballot = __ballot(hit);
      if (ballot) {
        const int lane = threadIdx.x & 63;
        unsigned basepos = 0;
        if (lane == 0) basepos = atomicAdd(&ctl->n_cand, (unsigned)__popcll(ballot));
        basepos = __shfl(basepos, 0);
        if (hit) {
          unsigned pos = basepos + (unsigned)__popcll(ballot & ((1ull << lane) - 1ull));
          if (pos < capacity) out[pos] = Cand{z, x, y, diff};
          else ctl->overflow = 1;
        }
      }
    }
  }
}


// ==== lazy background filter ===================================================================================
// The background filter (sigma 7.5: 61 taps per axis) is by far the most expensive part of get_seeds, but min_im
// enters the result only at voxels that are 3x3x3 maxima of max_im with max_im - min_im above the lowest threshold
// level — a few thousand voxels in a 2 x 10^8 voxel stack — and at their 26 neighbours (the not-a-local-minimum
// test).  So only the first (axis-0) pass is run on the whole stack.  From its output zp a rigorous lower bound of
// min_im follows without the other two passes: every later pass is a correlation with non-negative taps summing to 1,
// re-quantised monotonically, so min_im(z, x, y) >= min of zp(z, ., .) over the (x +- R, y +- R) window (reflected
// indices fall inside the clipped window), minus the rounding slack of two passes.  The window minimum is taken from
// per-plane B x B block minima (B >= R: the window touches at most 3 x 3 blocks).  Voxels that pass the local-maximum
// test, the edge test and `max_im - bound >= lowest level` are the candidates; for each of them one wave evaluates the
// axis-1 pass on the nine (z', x') rows it needs and the axis-2 pass at the 27 neighbourhood positions with exactly
// the dense kernels' arithmetic (NI_Correlate1D order, unfused multiply and add, re-quantisation after each axis), and
// applies the reference's tests to the exact values.  Seeds are therefore identical to the dense path bit for bit;
// the bound only decides where the exact computation happens.
struct Cand0 { int z, x, y; float cmax; };   // local maximum of max_im that passed the bound test

__device__ __forceinline__ int reflect_idx(int q, int n, int mode) {   // scipy 'reflect' / 'nearest' (gauss.hip border_idx)
  if (mode == IA3_MODE_NEAREST) return q < 0 ? 0 : (q >= n ? n - 1 : q);
  if (q >= 0 && q < n) return q;
  int p = 2 * n;
  q %= p;
  if (q < 0) q += p;
  return q < n ? q : p - 1 - q;
}

// per plane z and B x B block: minimum and maximum magnitude of zp.  One 256-thread block reads B rows of 256 * V
// consecutive y, V voxels (16 bytes when the rows allow it) per lane and row; every thread reduces its V columns over
// the rows, then groups of B / V lanes reduce across y.
template <class T, int B, int V>
__global__ __launch_bounds__(256) void blockmin_k(const T* __restrict__ zp, int Z, int X, int Y, int nbx, int nby,
                                                  float* __restrict__ bmin, float* __restrict__ babs) {
  const int y = (blockIdx.x * 256 + threadIdx.x) * V;
  const int bx = blockIdx.y, z = blockIdx.z;
  const int x0 = bx * B, x1 = x0 + B < X ? x0 + B : X;
  float m = INFINITY, a = 0.f;
  if (y < Y) {   // V > 1 only when Y % V == 0: a lane's V voxels are all inside
    const T* p = zp + ((size_t)z * X + x0) * Y + y;
    constexpr int U = 8;   // rows in flight per thread (the reduction is latency-bound otherwise)
    int x = x0;
    for (; x + U <= x1; x += U, p += (size_t)U * Y) {
      alignas(16) T v[U][V];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if constexpr (V * sizeof(T) == 16) *reinterpret_cast<uint4*>(v[u]) = *reinterpret_cast<const uint4*>(p + (size_t)u * Y);
        else v[u][0] = p[(size_t)u * Y];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int k = 0; k < V; ++k) { const float f = (float)v[u][k]; m = fminf(m, f); a = fmaxf(a, fabsf(f)); }
    }
    for (; x < x1; ++x, p += Y) {
      alignas(16) T v[V];
      if constexpr (V * sizeof(T) == 16) *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(p);
      else v[0] = *p;
#pragma unroll
      for (int k = 0; k < V; ++k) { const float f = (float)v[k]; m = fminf(m, f); a = fmaxf(a, fabsf(f)); }
    }
  }
#pragma unroll
  for (int o = B / V / 2; o >= 1; o >>= 1) {
    m = fminf(m, __shfl_xor(m, o)); a = fmaxf(a, __shfl_xor(a, o));
  }
  if ((threadIdx.x & (B / V - 1)) == 0 && y < Y) {
    const size_t o = ((size_t)z * nbx + bx) * nby + y / B;
    bmin[o] = m; babs[o] = a;
  }
}

// lower bound of min_im over each block: minimum over the 3 x 3 block neighbourhood minus the rounding slack of the two
// remaining passes (float32: a relative 2e-6 of the largest magnitude involved; uint16: each pass truncates, so up to
// one count per pass)
__global__ __launch_bounds__(256) void blockbound_k(const float* __restrict__ bmin, const float* __restrict__ babs, int Z,
                                                    int nbx, int nby, int is_u16, double* __restrict__ lb) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t n = (size_t)Z * nbx * nby;
  if (i >= n) return;
  const int by = (int)(i % nby), bx = (int)((i / nby) % nbx);
  const size_t pz = i - (size_t)bx * nby - by;
  float m = INFINITY, a = 0.f;
  for (int dx = -1; dx <= 1; ++dx)
    for (int dy = -1; dy <= 1; ++dy) {
      const int xx = bx + dx, yy = by + dy;
      if (xx < 0 || xx >= nbx || yy < 0 || yy >= nby) continue;
      const float v = bmin[pz + (size_t)xx * nby + yy], w = babs[pz + (size_t)xx * nby + yy];
      m = v < m ? v : m; a = w > a ? w : a;
    }
  lb[i] = (double)m - 2e-6 * (double)a - (is_u16 ? 2.0 : 0.0);
}

// The same bound from the column kernel's strip minima (gauss.hip: smallest value / largest magnitude of the axis-0 result per
// group of planes, row and 32-column strip, taken from registers — no pass over the stored stack): minimum over the rows of
// the 3 x 3 block neighbourhood, one value per plane group, written for every plane of the group.  Coarser along z than the
// per-plane block minima (the axis-0 result is smooth along z by construction); the extra uint16 count covers outputs the
// column kernel recomputes after taking their minimum (they may differ from the stored value by one count / one ulp).
__global__ __launch_bounds__(256) void stripbound_k(const float* __restrict__ smin, const float* __restrict__ sabs, int Z, int X,
                                                    int nbx, int nby, int ngz, int is_u16, double* __restrict__ lb) {
  // one wave per (plane group, block): its lanes share the <= 96 rows x 3 strips of the neighbourhood
  const size_t i = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= (size_t)ngz * nbx * nby) return;   // whole wave
  const int lane = threadIdx.x & 63;
  const int by = (int)(i % nby), bx = (int)((i / nby) % nbx), g = (int)(i / ((size_t)nbx * nby));
  const int x0 = max(32 * (bx - 1), 0), x1 = min(32 * (bx + 2), X), y0 = max(by - 1, 0), ny = min(by + 1, nby - 1) - y0 + 1;
  float m = INFINITY, a = 0.f;
  for (int e = lane; e < (x1 - x0) * ny; e += 64) {
    const size_t o = ((size_t)g * X + x0 + e / ny) * nby + y0 + e % ny;
    m = fminf(m, smin[o]); a = fmaxf(a, sabs[o]);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { m = fminf(m, __shfl_xor(m, o)); a = fmaxf(a, __shfl_xor(a, o)); }
  const double v = (double)m - 2e-6 * (double)a - (is_u16 ? 3.0 : 0.0);
  for (int z = lane; z < Z; z += 64)
    if (z * ngz / Z == g) lb[((size_t)z * nbx + bx) * nby + by] = v;
}

// 3x3x3 local maxima of max_im that pass the edge test and the bound test -> Cand0 list.  Same tiling as
// seed_detect3_tiled (16 x 64 tile + halo in LDS, double-buffered, rolling three-plane pipeline along z), one stack.
template <class T, int ZC, int B>
__global__ __launch_bounds__(256) void seed_cand3_tiled(const T* __restrict__ mx, const double* __restrict__ lb, int nbx, int nby,
                                                        int Z, int X, int Y, int edge, double th_test,
                                                        Cand0* __restrict__ out, unsigned capacity,
                                                        SeedCtl* __restrict__ ctl,
                                                        const float* __restrict__ smax, int sm_ty, int sm_ntile) {
  constexpr int TX = 16, TY = 64, HX = TX + 2, HY = TY + 2, NE = (HX * HY + 255) / 256;
  static_assert(B % TX == 0, "a tile must lie inside one block row");
  static_assert(ZC <= 254, "plane flags");
  __shared__ T tmax[2][HX][HY + 2];
  __shared__ unsigned char act_s[ZC + 2];   // act_s[1 + z - z0]: plane z of this tile can hold a candidate
  const int nty = (Y + TY - 1) / TY, ntx = (X + TX - 1) / TX;
  const int tile = xcd_tile(blockIdx.x, nty * ntx);
  if (tile < 0) return;
  const int x0 = (tile / nty) * TX, y0 = (tile % nty) * TY;
  const int z0 = blockIdx.z * ZC, z1 = z0 + ZC < Z ? z0 + ZC : Z;
  const int ty = threadIdx.x & 63, tg = threadIdx.x >> 6;
  size_t goff[NE]; int lr[NE], lc[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = threadIdx.x + 256 * i;
    const int r = e / HY, c = e % HY;
    lr[i] = e < HX * HY ? r : -1; lc[i] = c;
    const int gx = min(max(x0 + r - 1, 0), X - 1), gy = min(max(y0 + c - 1, 0), Y - 1);
    goff[i] = (size_t)gx * Y + gy;
  }
  T ra[NE];
  auto fetch = [&](int q) {
    const int zq = q < 0 ? 0 : (q >= Z ? Z - 1 : q);
    const size_t pz = (size_t)zq * X * Y;
#pragma unroll
    for (int i = 0; i < NE; ++i) if (lr[i] >= 0) ra[i] = mx[pz + goff[i]];
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NE; ++i) if (lr[i] >= 0) tmax[buf][lr[i]][lc[i]] = ra[i];
  };
  T pM[4][3], cM[4][2];
#pragma unroll
  for (int v = 0; v < 4; ++v) { pM[v][0] = pM[v][1] = pM[v][2] = 0; cM[v][0] = cM[v][1] = 0; }
  const int yb = min(y0 + ty, Y - 1) / B;
  const size_t lrow = (size_t)(x0 / B) * nby + yb;
  // Plane flags.  smax (optional, written by the plane-wise filter kernel): largest max_im value per plane and 16 x 64 tile
  // (this kernel's tiles); a plane of this tile whose maximum stays below `lowest level + smallest bound` cannot hold a
  // candidate (float subtraction is monotone, so the test below never drops a voxel the per-voxel test would pass).  Planes
  // that are neither flagged nor next to a flagged plane are not fetched at all.
  if (threadIdx.x < ZC + 2) {
    const int z = z0 - 1 + (int)threadIdx.x;
    unsigned char a = 0;
    if (z >= z0 && z < z1) {
      a = 1;
      if (smax) {
        const int ylast = min(y0 + TY - 1, Y - 1);
        const float m = smax[((size_t)z * ((X + 15) / 16) + x0 / 16) * ((Y + 63) / 64) + y0 / 64];   // this tile's own entry
        double lo = INFINITY;
        for (int by = y0 / B; by <= ylast / B; ++by) lo = fmin(lo, lb[(size_t)z * nbx * nby + (size_t)(x0 / B) * nby + by]);
        a = ((double)m - lo >= th_test) ? 1 : 0;
      }
    }
    act_s[threadIdx.x] = a;
  }
  __syncthreads();
  auto act = [&](int z) -> bool { return act_s[1 + z - z0] != 0; };                       // z in [z0 - 1, z1]
  auto need = [&](int q) -> bool {                                                          // q in [z0 - 1, z1]
    return (q > z0 - 1 && act(q - 1)) || act(q) || (q < z1 && act(q + 1));
  };
  if (need(z0 - 1)) { fetch(z0 - 1); stash(0); }
  __syncthreads();
  for (int q = z0 - 1, buf = 0; q <= z1; ++q, buf ^= 1) {
    const bool nq = need(q), nn = q < z1 && need(q + 1);   // block-uniform
    if (nn) fetch(q + 1);
    if (nq) {
      const int c = ty + 1;
      T hM[6], ce[4];
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) {
        const int r = tg * 4 + rr;
        const T a0 = tmax[buf][r][c - 1], a1 = tmax[buf][r][c], a2 = tmax[buf][r][c + 1];
        hM[rr] = max3v(a0, a1, a2);
        if (rr >= 1 && rr <= 4) ce[rr - 1] = a1;
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const T a = max3v(hM[v], hM[v + 1], hM[v + 2]);
        pM[v][0] = pM[v][1]; pM[v][1] = pM[v][2]; pM[v][2] = a;
        cM[v][0] = cM[v][1]; cM[v][1] = ce[v];
      }
    }
    if (nn) stash(buf ^ 1);
    if (nq || nn) __syncthreads();   // this plane's reads before the plane after next is stashed over them; the next plane's stash
    const int z = q - 1;
    if (z < z0 || !act(z)) continue;
    const double bound = lb[(size_t)z * nbx * nby + lrow];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int x = x0 + tg * 4 + v, y = y0 + ty;
      const T vmax = max3v(pM[v][0], pM[v][1], pM[v][2]);
      const T cmax = cM[v][0];
      bool hit = x < X && y < Y && (vmax == cmax) && ((double)cmax - bound >= th_test);
      if (edge > 0)
        hit = hit && z >= edge && z <= Z - edge && x >= edge && x <= X - edge && y >= edge && y <= Y - edge;
      const unsigned long long ballot = __ballot(hit);
      if (ballot) {
        const int lane = threadIdx.x & 63;
        unsigned basepos = 0;
        if (lane == 0) basepos = atomicAdd(&ctl->n_cand, (unsigned)__popcll(ballot));
        basepos = __shfl(basepos, 0);
        if (hit) {
          unsigned pos = basepos + (unsigned)__popcll(ballot & ((1ull << lane) - 1ull));
          if (pos < capacity) out[pos] = Cand0{z, x, y, (float)cmax};
          else ctl->overflow = 1;
        }
      }
    }
  }
}

template <class T> __device__ __forceinline__ float quant(double v);
template <> __device__ __forceinline__ float quant<float>(double v) { return (float)v; }
template <> __device__ __forceinline__ float quant<uint16_t>(double v) { return (float)(uint16_t)(int)v; }

struct TapsD { double w[64]; };   // w[j] = tap at offset j (symmetric), j <= R

// One 576-thread block per candidate (grid-stride), one wave per row: exact min_im at the candidate and its 26
// neighbours from the axis-0 result zp.
//   rows r = 3 * dz + dx (z' = clamp(z + dz - 1), x' = clamp(x + dx - 1)), positions i = 0 .. 2R+2 <-> y'' = y - 1 - R + i
//   (reflected): trow[r][i] = axis-1 pass of zp at (z', x', y''), quantised to the stack dtype (lanes run along i, so
//   every load is a contiguous row piece; the nine rows go side by side because the pass is a chain of dependent
//   loads); then lane k < 27 of wave 0 runs the axis-2 pass at (row k / 3, y' = clamp(y + k % 3 - 1)) over its row in
//   LDS.  Both passes: acc = in[0] * w0; for j = R..1: acc = acc + (in[-j] + in[+j]) * w[j]  (this file is compiled
//   with -ffp-contract=off), as NI_Correlate1D and the dense kernels do.
template <class T>
__global__ __launch_bounds__(576) void bg_sparse_k(const T* __restrict__ zp, int Z, int X, int Y, TapsD taps, int R, int mode,
                                                   const Cand0* __restrict__ c0, const SeedCtl* __restrict__ ctl0,
                                                   unsigned cap0, double th_low, Cand* __restrict__ out, unsigned capacity,
                                                   SeedCtl* __restrict__ ctl) {
  constexpr int NI = 2 * 63 + 3;              // positions per row at the largest supported radius
  __shared__ float trow[9][NI + 1];
  __shared__ float mval[32];
  const int r = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned n0 = ctl0->n_cand < cap0 ? ctl0->n_cand : cap0;
  const int ni = 2 * R + 3;
  for (unsigned c = blockIdx.x; c < n0; c += gridDim.x) {   // block-uniform
    const Cand0 k = c0[c];
    {   // ---- axis 1 on row r ----
      const int zz = min(max(k.z + r / 3 - 1, 0), Z - 1), xx = min(max(k.x + r % 3 - 1, 0), X - 1);
      const T* pl = zp + (size_t)zz * X * Y;
      for (int i = lane; i < ni; i += 64) {
        const int yy = reflect_idx(k.y - 1 - R + i, Y, mode);
        double acc = (double)pl[(size_t)xx * Y + yy] * taps.w[0];
#pragma unroll 8
        for (int j = R; j >= 1; --j) {
          const double a = (double)pl[(size_t)reflect_idx(xx - j, X, mode) * Y + yy];
          const double b = (double)pl[(size_t)reflect_idx(xx + j, X, mode) * Y + yy];
          acc = acc + (a + b) * taps.w[j];
        }
        trow[r][i] = quant<T>(acc);
      }
    }
    __syncthreads();
    // ---- axis 2 at the 27 neighbourhood positions ----
    if (threadIdx.x < 27) {
      const int rr = threadIdx.x / 3;
      const int yc = min(max(k.y + (int)threadIdx.x % 3 - 1, 0), Y - 1);
      const int i0 = yc - (k.y - 1 - R);        // in [R, R + 2]
      const float* t = trow[rr];
      // reflect(yc + j) = reflect(y - 1 - R + (i0 + j)): the row array already holds the reflected positions
      double acc = (double)t[i0] * taps.w[0];
      for (int j = R; j >= 1; --j) acc = acc + ((double)t[i0 - j] + (double)t[i0 + j]) * taps.w[j];
      mval[threadIdx.x] = quant<T>(acc);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const float cmin = mval[13];
      float vmin = cmin;
      for (int q = 0; q < 27; ++q) vmin = mval[q] < vmin ? mval[q] : vmin;
      const float diff = k.cmax - cmin;                       // float32(max_im) - float32(min_im), fitting.py:106
      if (vmin != cmin && (double)diff >= th_low) {
        const unsigned pos = atomicAdd(&ctl->n_cand, 1u);
        if (pos < capacity) out[pos] = Cand{k.z, k.x, k.y, diff};
        else ctl->overflow = 1;
      }
    }
    __syncthreads();   // the next candidate's rows overwrite trow
  }
}

// The same computation for the production radius (RC = 30 at compile time) with one WAVE per plane z' and three waves per
// candidate.  The three rows x' = x - 1, x, x + 1 of a plane take their 61 axis-1 taps from 63 consecutive rows of the
// stack: a lane loads those 63 values once (all in flight together, row starts in scalar registers) and runs the three
// chains on registers, instead of 3 x 61 loads in eight dependent batches.  Candidates whose 63 rows touch the border
// of the stack (reflection, clamped x') take the row-by-row code of bg_sparse_k.  Same operations in the same order.
template <class T, int RC>
__global__ __launch_bounds__(192) void bg_sparse3_k(const T* __restrict__ zp, int Z, int X, int Y, TapsD taps, int mode,
                                                    const Cand0* __restrict__ c0, const SeedCtl* __restrict__ ctl0,
                                                    unsigned cap0, double th_low, Cand* __restrict__ out, unsigned capacity,
                                                    SeedCtl* __restrict__ ctl) {
  constexpr int NI = 2 * RC + 3;              // positions per row; also the rows a lane loads
  static_assert(NI <= 64, "one lane per position");
  __shared__ float trow[9][NI + 1];
  __shared__ float mval[32];
  const int dzw = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned n0 = ctl0->n_cand < cap0 ? ctl0->n_cand : cap0;
  for (unsigned c = blockIdx.x; c < n0; c += gridDim.x) {   // block-uniform
    const Cand0 k = c0[c];
    const int kz = __builtin_amdgcn_readfirstlane(k.z), kx = __builtin_amdgcn_readfirstlane(k.x),
              ky = __builtin_amdgcn_readfirstlane(k.y);
    const int zz = min(max(kz + dzw - 1, 0), Z - 1);
    const T* pl = zp + (size_t)zz * X * Y;
    if (kx - 1 - RC >= 0 && kx + 1 + RC < X) {   // block-uniform
      if (lane < NI) {
        const unsigned yy = (unsigned)reflect_idx(ky - 1 - RC + lane, Y, mode);
        const T* base = pl + (size_t)(kx - 1 - RC) * Y;
        T v[NI];
#pragma unroll
        for (int m = 0; m < NI; ++m) v[m] = (base + (size_t)m * Y)[yy];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          double acc = (double)v[RC + dx] * taps.w[0];
#pragma unroll
          for (int j = RC; j >= 1; --j) acc = acc + ((double)v[RC + dx - j] + (double)v[RC + dx + j]) * taps.w[j];
          trow[3 * dzw + dx][lane] = quant<T>(acc);
        }
      }
    } else {
      for (int dx = 0; dx < 3; ++dx) {
        const int xx = min(max(kx + dx - 1, 0), X - 1);
        if (lane < NI) {
          const int yy = reflect_idx(ky - 1 - RC + lane, Y, mode);
          double acc = (double)pl[(size_t)xx * Y + yy] * taps.w[0];
#pragma unroll 8
          for (int j = RC; j >= 1; --j) {
            const double a = (double)pl[(size_t)reflect_idx(xx - j, X, mode) * Y + yy];
            const double b = (double)pl[(size_t)reflect_idx(xx + j, X, mode) * Y + yy];
            acc = acc + (a + b) * taps.w[j];
          }
          trow[3 * dzw + dx][lane] = quant<T>(acc);
        }
      }
    }
    __syncthreads();
    // ---- axis 2 at the 27 neighbourhood positions ----
    if (threadIdx.x < 27) {
      const int rr = threadIdx.x / 3;
      const int yc = min(max(ky + (int)threadIdx.x % 3 - 1, 0), Y - 1);
      const int i0 = yc - (ky - 1 - RC);        // in [RC, RC + 2]
      const float* t = trow[rr];
      float ta[RC], tb[RC];
#pragma unroll
      for (int j = 1; j <= RC; ++j) { ta[j - 1] = t[i0 - j]; tb[j - 1] = t[i0 + j]; }
      double acc = (double)t[i0] * taps.w[0];
#pragma unroll
      for (int j = RC; j >= 1; --j) acc = acc + ((double)ta[j - 1] + (double)tb[j - 1]) * taps.w[j];
      mval[threadIdx.x] = quant<T>(acc);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const float cmin = mval[13];
      float vmin = cmin;
      for (int q = 0; q < 27; ++q) vmin = mval[q] < vmin ? mval[q] : vmin;
      const float diff = k.cmax - cmin;                       // float32(max_im) - float32(min_im), fitting.py:106
      if (vmin != cmin && (double)diff >= th_low) {
        const unsigned pos = atomicAdd(&ctl->n_cand, 1u);
        if (pos < capacity) out[pos] = Cand{k.z, k.x, k.y, diff};
        else ctl->overflow = 1;
      }
    }
    __syncthreads();   // the next candidate's rows overwrite trow
  }
}

template <class T>
void launch_detect(int W, const void* mx, const void* mn, int Z, int X, int Y, int edge, double th_low,
                   Cand* out, unsigned capacity, SeedCtl* ctl, hipStream_t s, int rule = 0) {
  if (W == 3 && rule == 0) {
    constexpr int ZT = 64;   // planes per block; the two halo planes of a chunk are re-read
    const unsigned tiles = (unsigned)((Y + 63) / 64) * (unsigned)((X + 15) / 16);
    dim3 gt(8 * ((tiles + 7) / 8), 1, (unsigned)((Z + ZT - 1) / ZT));
    hipLaunchKernelGGL((seed_detect3_tiled<T, ZT>), gt, dim3(256), 0, s, (const T*)mx, (const T*)mn, Z, X, Y, edge, th_low,
                       out, capacity, ctl);
    return;
  }
  constexpr int ZC = 10;
  dim3 g((unsigned)((Y + 63) / 64), (unsigned)((X + 3) / 4), (unsigned)((Z + ZC - 1) / ZC));
#define IA3_SEED_CASE(WW)                                                                              \
  case WW:                                                                                             \
    hipLaunchKernelGGL((seed_detect<T, ZC, WW>), g, dim3(256), 0, s, (const T*)mx, (const T*)mn, Z, X, Y, \
                       edge, th_low, out, capacity, ctl, rule);                                        \
    break;
  switch (W) {
    IA3_SEED_CASE(1) IA3_SEED_CASE(2) IA3_SEED_CASE(3) IA3_SEED_CASE(4)
    IA3_SEED_CASE(5) IA3_SEED_CASE(6) IA3_SEED_CASE(7) IA3_SEED_CASE(8)
  }
#undef IA3_SEED_CASE
}

}  // namespace

namespace {

// ---- tail of get_seeds on the device (fitting.py:113-150) ---------------------------------------------------------
// level pick, hot-column vote, sort by height and truncation as four small kernels: one block picks the level and packs
// the candidates that pass it (a corrected production image leaves ~13 000 at the lowest dynamic level and ~5 000 at the
// one that is taken), then all-pairs work over those (n = 5 000: 25 M comparisons spread over 160 blocks; unique 64-bit
// keys => rank = position): no sort network, no hash table, and the seed list never leaves HBM between the detector and
// the fitter.  More candidates than FIN_CAP, or more than FIN_KEYS at the chosen level: the host's finish takes over.
// (Round 4 also built the whole tail as ONE single-block kernel — keys in 128 KB of LDS, two bitonic sorts — and measured
// it at 163-173 us against 46-60 us for these four launches: 182 compare-exchange stages of a 16-wave block, each with
// its barrier, cost ~0.9 us apiece.)
constexpr unsigned FIN_CAP = 32768;    // candidates the first kernel looks at
constexpr unsigned FIN_KEYS = 16384;   // ... of which at most this many may pass the chosen level
constexpr int FIN_S = 8;               // slices of the all-pairs loops (grid.y)
struct FinCtl { unsigned n_alive; int chosen; unsigned n_cand, overflow; unsigned done, n_kept, pad[2]; };   // n_cand / overflow mirror SeedCtl

__device__ __forceinline__ unsigned fin_n(const SeedCtl* sctl) { return sctl->n_cand < FIN_CAP ? sctl->n_cand : FIN_CAP; }
__device__ __forceinline__ unsigned long long fin_key(const Cand& k) {   // h desc, then z, x, y desc (finish_seeds)
  uint32_t u = __float_as_uint(k.h);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)u << 32) | ((unsigned long long)k.z << 24) | ((unsigned long long)k.x << 12) | (unsigned long long)k.y;
}

// one block: dynamic threshold (first level whose count reaches min_dynamic_seeds, else the last one, :113-125), the
// candidates that pass it packed into kc[] (any order), the work arrays of the next kernels cleared
__global__ __launch_bounds__(1024) void fin_levels_k(const SeedCtl* __restrict__ sctl, const Cand* __restrict__ c, Levels lev,
                                                     int min_dyn, FinCtl* __restrict__ fc, Cand* __restrict__ kc,
                                                     unsigned* __restrict__ hotcnt, unsigned* __restrict__ rank) {
  __shared__ unsigned cnt[MAXLEV];
  __shared__ unsigned nk;
  __shared__ int chosen_s;
  const unsigned tid = threadIdx.x;
  if (tid < MAXLEV) cnt[tid] = 0;
  if (tid == 0) nk = 0;
  __syncthreads();
  const unsigned n = fin_n(sctl);
  for (unsigned i = tid; i < n; i += 1024) {
    const double h = (double)c[i].h;
    for (int l = 0; l < lev.n; ++l) if (h >= lev.th[l]) atomicAdd(&cnt[l], 1u);
  }
  __syncthreads();
  if (tid == 0) {
    int chosen = lev.n - 1;
    for (int l = 0; l < lev.n; ++l) if ((long long)cnt[l] >= (long long)min_dyn) { chosen = l; break; }
    chosen_s = chosen;
  }
  __syncthreads();
  const double th = lev.th[chosen_s];
  for (unsigned i0 = 0; i0 < n; i0 += 1024) {   // wave-aggregated append
    const unsigned i = i0 + tid;
    Cand k{0, 0, 0, 0.f};
    if (i < n) k = c[i];
    const bool keep = i < n && (double)k.h >= th;
    const unsigned long long m = __ballot(keep);
    unsigned base = 0;
    if ((tid & 63) == 0 && m) base = atomicAdd(&nk, (unsigned)__popcll(m));
    base = __shfl(base, 0, 64);
    const unsigned pos = base + (unsigned)__popcll(m & ((1ull << (tid & 63)) - 1ull));
    if (keep && pos < FIN_KEYS) kc[pos] = k;
  }
  __syncthreads();
  const unsigned kept = nk < FIN_KEYS ? nk : FIN_KEYS;
  for (unsigned i = tid; i < kept; i += 1024) { hotcnt[i] = 0u; rank[i] = 0u; }
  if (tid == 0) { fc->chosen = chosen_s; fc->n_kept = nk; fc->n_alive = 0u; fc->done = 0u; }
}
__device__ __forceinline__ unsigned fin_kept(const FinCtl* fc) { return fc->n_kept <= FIN_KEYS ? fc->n_kept : 0u; }   // too many: nothing to do

// hotcnt[i] = number of packed candidates in the (x, y) column of packed candidate i
__global__ __launch_bounds__(256) void fin_hot_k(const Cand* __restrict__ kc, const FinCtl* __restrict__ fc, unsigned* __restrict__ hotcnt) {
  __shared__ int tx[256], ty[256];
  const unsigned n = fin_kept(fc);
  if (blockIdx.x * 256 >= n) return;   // whole block
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  const int xi = i < n ? kc[i].x : -1, yi = i < n ? kc[i].y : -1;
  const unsigned per = (n + FIN_S - 1) / FIN_S, j0 = blockIdx.y * per, j1 = j0 + per < n ? j0 + per : n;
  unsigned cnt = 0;
  for (unsigned jb = j0; jb < j1; jb += 256) {
    const unsigned j = jb + threadIdx.x;
    __syncthreads();
    tx[threadIdx.x] = j < j1 ? kc[j].x : -2;
    ty[threadIdx.x] = j < j1 ? kc[j].y : -2;
    __syncthreads();
    const unsigned m = j1 - jb < 256 ? j1 - jb : 256;
    for (unsigned t = 0; t < m; ++t) cnt += (tx[t] == xi && ty[t] == yi);
  }
  if (i < n && cnt) atomicAdd(&hotcnt[i], cnt);
}

// rank[i] = number of surviving candidates that sort before packed candidate i; fc->n_alive = survivors
__global__ __launch_bounds__(256) void fin_rank_k(const Cand* __restrict__ kc, FinCtl* __restrict__ fc, const unsigned* __restrict__ hotcnt,
                                                  int hot_th, unsigned* __restrict__ rank) {
  __shared__ unsigned long long tk[256];
  const unsigned n = fin_kept(fc);
  if (blockIdx.x * 256 >= n) return;   // whole block
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  const bool alive = i < n && (hot_th <= 0 || hotcnt[i] < (unsigned)hot_th);
  const unsigned long long ki = i < n ? fin_key(kc[i]) : ~0ull;
  const unsigned per = (n + FIN_S - 1) / FIN_S, j0 = blockIdx.y * per, j1 = j0 + per < n ? j0 + per : n;
  unsigned r = 0;
  for (unsigned jb = j0; jb < j1; jb += 256) {
    const unsigned j = jb + threadIdx.x;
    __syncthreads();
    const bool aj = j < j1 && (hot_th <= 0 || hotcnt[j] < (unsigned)hot_th);
    tk[threadIdx.x] = aj ? fin_key(kc[j]) : 0ull;   // 0 sorts after every real key (a real key has bit 63 or a positive h)
    __syncthreads();
    const unsigned m = j1 - jb < 256 ? j1 - jb : 256;
    for (unsigned t = 0; t < m; ++t) r += tk[t] > ki;
  }
  if (alive) {
    if (r) atomicAdd(&rank[i], r);
    if (blockIdx.y == 0) atomicAdd(&fc->n_alive, 1u);
  }
}

__global__ __launch_bounds__(256) void fin_scatter_k(const SeedCtl* __restrict__ sctl, const Cand* __restrict__ kc, const FinCtl* fc,
                                                     const unsigned* __restrict__ hotcnt, int hot_th, const unsigned* __restrict__ rank,
                                                     int max_num, double* __restrict__ zxy, double* __restrict__ hh, FinCtl* fcw,
                                                     const SeedCtl* __restrict__ lazy, unsigned cap0,
                                                     volatile unsigned* __restrict__ mail, unsigned seq) {
  const unsigned n = fin_kept(fc);
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const bool alive = hot_th <= 0 || hotcnt[i] < (unsigned)hot_th;
    const unsigned r = alive ? rank[i] : 0u;
    if (alive && !(max_num > 0 && r >= (unsigned)max_num)) {
      const Cand k = kc[i];
      zxy[3 * r] = k.z; zxy[3 * r + 1] = k.x; zxy[3 * r + 2] = k.y;
      hh[r] = (double)k.h;
    }
  }
  // The host polls the mailbox and may hand the seed list to ANOTHER stream (the group fitter of ia3_fit_fovs) as soon as
  // it sees the sequence number, before this kernel has ended: the word is therefore published by the block that
  // finishes LAST, after every block's seeds have been released to device scope.
  __threadfence();
  __syncthreads();
  if (threadIdx.x != 0) return;
  if (atomicAdd(&fcw->done, 1u) != gridDim.x - 1) return;
  __threadfence();
  // overflow bit 1: the lazy path's first-stage list overflowed (the caller falls back to the dense filter); bit 2: more
  // candidates at the chosen level than FIN_KEYS (the host's finish takes over; n_alive is 0 then)
  const unsigned nc = sctl->n_cand;
  const unsigned ov = sctl->overflow | ((lazy && (lazy->overflow || lazy->n_cand > cap0)) ? 2u : 0u) | (fc->n_kept > FIN_KEYS ? 4u : 0u);
  fcw->n_cand = nc;
  fcw->overflow = ov;
  if (mail) {   // the four control words straight into the host's pinned mailbox, then the sequence number it polls
    mail[1] = fcw->n_alive; mail[2] = (unsigned)fcw->chosen; mail[3] = nc; mail[4] = ov;
    __threadfence_system();
    mail[0] = seq;
  }
}

}  // namespace

using namespace ia3rt;

namespace ia3k {

// Host tail of get_seeds: level pick, hot columns, sort, truncate (fitting.py:113-150) on a few thousand records.
static void finish_seeds(std::vector<Cand>& c, const Levels& lev, const ia3_seed_params& p, int Y,
                         SeedOut& o) {
  // dynamic threshold: first level whose (edge-filtered) count reaches min_dynamic_seeds,
  // else the last level (fitting.py:113-125: the loop variable keeps its last value)
  int chosen = lev.n - 1;
  for (int i = 0; i < lev.n; ++i) {
    long long cnt = 0;
    for (auto& k : c) cnt += ((double)k.h >= lev.th[i]);
    if (cnt >= (long long)p.min_dynamic_seeds) { chosen = i; break; }
  }
  o.th_used = lev.th[chosen];
  std::vector<Cand> s;
  s.reserve(c.size());
  for (auto& k : c) if ((double)k.h >= lev.th[chosen]) s.push_back(k);
  if (p.remove_hot_pixel && !s.empty()) {  // fitting.py:131-138: drop (x,y) seen in >= hot_pixel_th planes
    // open-addressing count table over the (x,y) keys
    size_t cap = 64;
    while (cap < 4 * s.size()) cap <<= 1;
    std::vector<long long> keys(cap, -1);
    std::vector<int> cnts(cap, 0);
    auto slot = [&](long long key) {
      size_t h = (size_t)((unsigned long long)key * 0x9E3779B97F4A7C15ull >> 20) & (cap - 1);
      while (keys[h] != -1 && keys[h] != key) h = (h + 1) & (cap - 1);
      return h;
    };
    bool any_hot = false;
    for (auto& k : s) {
      const long long key = (long long)k.x * (Y + 1) + k.y;
      size_t h = slot(key);
      keys[h] = key;
      if (++cnts[h] >= p.hot_pixel_th) any_hot = true;
    }
    if (any_hot) {
      std::vector<Cand> kept;
      kept.reserve(s.size());
      for (auto& k : s) if (cnts[slot((long long)k.x * (Y + 1) + k.y)] < p.hot_pixel_th) kept.push_back(k);
      s.swap(kept);
    }
  }
  // np.flipud(np.argsort(h)) on the np.where-ordered list: ascending by h then reversed.  NumPy's quicksort is
  // not stable, so the order inside groups of equal h is implementation-defined there; here ties come out in
  // descending np.where (z, x, y) order = what a stable ascending sort followed by the flip gives.
  std::sort(s.begin(), s.end(), [](const Cand& a, const Cand& b) {
    if (a.h != b.h) return a.h > b.h;
    if (a.z != b.z) return a.z > b.z;
    if (a.x != b.x) return a.x > b.x;
    return a.y > b.y;
  });
  if (p.max_num_seeds > 0 && (size_t)p.max_num_seeds <= s.size()) s.resize(p.max_num_seeds);
  o.zxyh.resize(s.size() * 4);
  for (size_t i = 0; i < s.size(); ++i) {
    o.zxyh[4 * i + 0] = s[i].z; o.zxyh[4 * i + 1] = s[i].x; o.zxyh[4 * i + 2] = s[i].y; o.zxyh[4 * i + 3] = (double)s[i].h;
  }
}

static double now_ms() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

int g_strip_bound = 1;   // IA3_TUNE_SEED_STRIPS: 0 = block minima from a pass over the axis-0 result even where the column kernel supplies strips
void set_seed_strips(int on) { g_strip_bound = on ? 1 : 0; }
int g_seed_dense = 0;   // IA3_TUNE_SEED_DENSE: 1 = always run the dense background filter (tests compare the two paths)
void set_seed_dense(int on) { g_seed_dense = on ? 1 : 0; }

constexpr unsigned LAZY_CAP = 1u << 17;   // first-stage candidates of the lazy background path (2 MB)

// layout of the lazy filter's bound block: [nb float block minima | nb float block magnitudes | pad to 8 bytes | nb double
// lower bounds]; the block itself is 256-byte aligned (scratch cache), so the doubles are 8-byte aligned for every nb
inline size_t lazy_bound_offset(size_t nb) { return 2 * ((nb + 1) & ~(size_t)1); }   // in floats
inline double* lazy_bound_ptr(void* bnd, size_t nb) { return (double*)((float*)bnd + lazy_bound_offset(nb)); }
inline size_t lazy_bound_bytes(size_t nb) { return lazy_bound_offset(nb) * sizeof(float) + nb * sizeof(double); }

template <class T>
static void launch_lazy(const void* mx, const void* zp, int Z, int X, int Y, const double* w, int R, int edge,
                        double th_low, void* bnd, Cand0* c0, SeedCtl* ctl0, Cand* out, unsigned capacity, SeedCtl* ctl,
                        hipStream_t s, int stage, const float* smax = nullptr, int sm_ty = 0, int sm_ntile = 0) {
  const int B = R <= 32 ? 32 : 64;
  const int nbx = (X + B - 1) / B, nby = (Y + B - 1) / B;
  const size_t nb = (size_t)Z * nbx * nby;
  float* bmin = (float*)bnd;
  float* babs = bmin + nb;
  double* lb = lazy_bound_ptr(bnd, nb);
  if (stage == 0) {   // needs only the axis-0 result: queued before the front filter is joined
    ProfScope ps("seed_blockmin");
    constexpr int V = 16 / (int)sizeof(T);
    const bool wide = Y % V == 0 && ((uintptr_t)zp & 15) == 0;   // every row starts on a 16-byte boundary
    const int per = 256 * (wide ? V : 1);
    dim3 g((unsigned)((Y + per - 1) / per), (unsigned)nbx, (unsigned)Z);
    if (B == 32 && wide) hipLaunchKernelGGL((blockmin_k<T, 32, V>), g, dim3(256), 0, s, (const T*)zp, Z, X, Y, nbx, nby, bmin, babs);
    else if (B == 32) hipLaunchKernelGGL((blockmin_k<T, 32, 1>), g, dim3(256), 0, s, (const T*)zp, Z, X, Y, nbx, nby, bmin, babs);
    else if (wide) hipLaunchKernelGGL((blockmin_k<T, 64, V>), g, dim3(256), 0, s, (const T*)zp, Z, X, Y, nbx, nby, bmin, babs);
    else hipLaunchKernelGGL((blockmin_k<T, 64, 1>), g, dim3(256), 0, s, (const T*)zp, Z, X, Y, nbx, nby, bmin, babs);
    hipLaunchKernelGGL(blockbound_k, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, (const float*)bmin, (const float*)babs,
                       Z, nbx, nby, (int)(sizeof(T) == 2), lb);
    return;
  }
  {
    ProfScope ps("seed_detect");
    constexpr int ZT = 32;   // planes per block: two chunks of a 50-plane stack balance the skipped planes better than one (0.174 -> 0.164 ms)
    const unsigned tiles = (unsigned)((Y + 63) / 64) * (unsigned)((X + 15) / 16);
    dim3 gt(8 * ((tiles + 7) / 8), 1, (unsigned)((Z + ZT - 1) / ZT));
    const double th_test = th_low - fabs(th_low) * 1e-6 - 1e-300;   // the exact test is made in float32: keep the bound test looser
    if (B == 32)
      hipLaunchKernelGGL((seed_cand3_tiled<T, ZT, 32>), gt, dim3(256), 0, s, (const T*)mx, (const double*)lb, nbx, nby, Z, X, Y, edge,
                         th_test, c0, LAZY_CAP, ctl0, smax, sm_ty, sm_ntile);
    else
      hipLaunchKernelGGL((seed_cand3_tiled<T, ZT, 64>), gt, dim3(256), 0, s, (const T*)mx, (const double*)lb, nbx, nby, Z, X, Y, edge,
                         th_test, c0, LAZY_CAP, ctl0, smax, sm_ty, sm_ntile);
  }
  {
    ProfScope ps("seed_sparse_bg");
    TapsD t;
    for (int j = 0; j < 64; ++j) t.w[j] = j <= R ? w[R + j] : 0.0;
    if (R == 30)   // the production radius: 0.163 -> 0.128 ms for 5 000 candidates (grids of 1024 / 2048 / 8192 blocks: 0.158 / 0.134 / 0.129)
      hipLaunchKernelGGL((bg_sparse3_k<T, 30>), dim3(4096), dim3(192), 0, s, (const T*)zp, Z, X, Y, t, IA3_MODE_REFLECT,
                         (const Cand0*)c0, (const SeedCtl*)ctl0, LAZY_CAP, th_low, out, capacity, ctl);
    else
    hipLaunchKernelGGL((bg_sparse_k<T>), dim3(4096), dim3(576), 0, s, (const T*)zp, Z, X, Y, t, R, IA3_MODE_REFLECT,
                       (const Cand0*)c0, (const SeedCtl*)ctl0, LAZY_CAP, th_low, out, capacity, ctl);
  }
}

static int dog_seed_impl(const ia3_stack* im, const ia3_seed_params& p, SeedOut& out, SeedDev* dev, bool force_dense = false) {
  hipStream_t s = stream();
  const bool dbg = getenv("IA3_DEBUG_TIMING") != nullptr;
  const double t0 = now_ms();
  const int Z = im->Z, X = im->X, Y = im->Y;
  const size_t bytes = im->bytes;
  if (p.filt_size < 1 || p.filt_size > 8) return set_error(IA3_EUNSUPPORTED, "filt_size %d not in 1..8", p.filt_size);
  int niter = p.use_dynamic_th ? p.dynamic_niters : 1;
  if (niter < 1) niter = 1;  // range(0) would leave _coords undefined in the reference
  if (niter > MAXLEV) return set_error(IA3_EUNSUPPORTED, "dynamic_niters > %d", MAXLEV);
  Levels lev;
  lev.n = niter;
  for (int i = 0; i < niter; ++i) {
    double t = p.th_seed * (1 - (double)i / (double)niter);
    lev.th[i] = p.th_compare_f32 ? (double)(float)t : t;
  }
  double th_low = lev.th[0];
  for (int i = 1; i < lev.n; ++i) th_low = lev.th[i] < th_low ? lev.th[i] : th_low;
  // filtered stacks
  Scratch a(bytes), b(bytes), tmp(bytes), tmp2(bytes);   // tmp2: ping-pong buffer of the front filter (own stream)
  if (!a.p || !b.p || !tmp.p || !tmp2.p) return IA3_ENOMEM;
  const void* maxim = im->d;
  const void* minim = im->d;
  std::vector<double> w, wb; int R, Rb = 0, rc;
  if (p.background_gfilt_size > 0) {
    if (p.w_back) { wb.assign(p.w_back, p.w_back + 2 * p.r_back + 1); Rb = p.r_back; }
    else gaussian_taps(p.background_gfilt_size, 4.0, wb, Rb);
  }
  // lazy background filter (see the kernels above): axis 0 everywhere, axes 1 and 2 only around candidate maxima
  bool lazy = !force_dense && !g_seed_dense && p.filt_size == 3 && p.background_gfilt_size > 0 && Rb > 3 && Rb <= 63;
  if (lazy)
    for (int j = 0; j <= 2 * Rb; ++j)   // the bound needs non-negative taps (any Gaussian; explicit taps are checked)
      if (!(wb[j] >= 0.0) || wb[j] != wb[2 * Rb - j]) lazy = false;
  const int Bk = Rb <= 32 ? 32 : 64;
  const size_t nblk = (size_t)Z * ((X + Bk - 1) / Bk) * ((Y + Bk - 1) / Bk);
  Scratch bnd(lazy ? lazy_bound_bytes(nblk) : 256);
  Scratch c0buf(lazy ? (size_t)LAZY_CAP * sizeof(Cand0) : 256);
  if (!bnd.p || !c0buf.p) return IA3_ENOMEM;
  // candidate buffer = [SeedCtl out | SeedCtl lazy | Cand x capacity] and the workspace of the device-side finish; they are
  // cleared on the main stream right behind the first filter launch (clear_buffers below), where the fills run under the
  // other filter kernels: neither in front of the first kernel of the FOV nor between the filters and the detector
  unsigned capacity = 1u << 20;
  constexpr size_t HDR = 2 * sizeof(SeedCtl);
  Scratch buf0(HDR + (size_t)capacity * sizeof(Cand));
  if (!buf0.p) return IA3_ENOMEM;
  const bool dev_finish = dev && Z <= 256 && X <= 4096 && Y <= 4096;
  // [FinCtl | hot | rank | zxy | h | packed candidates]
  const size_t o_hot = 256, o_rank = o_hot + 4 * (size_t)FIN_KEYS, o_zxy = o_rank + 4 * (size_t)FIN_KEYS,
               o_h = o_zxy + 24 * (size_t)FIN_KEYS, o_kc = o_h + 8 * (size_t)FIN_KEYS, fin_bytes = o_kc + sizeof(Cand) * (size_t)FIN_KEYS;
  void* fin = nullptr;
  if (dev_finish) {
    fin = ws_get(fin_bytes);
    if (!fin) return IA3_ENOMEM;
  }
  struct FinGuard { void*& p; ~FinGuard() { if (p) ws_put(p); } } fin_guard{fin};   // handed to the caller on success (fin = nullptr)
  bool cleared = false;
  auto clear_buffers = [&]() -> int {
    if (cleared) return IA3_OK;
    cleared = true;
    const hipStream_t cs = stream();   // the auxiliary stream inside an AuxResume
    IA3_HIP(hipMemsetAsync(buf0.p, 0, HDR, cs));
    return IA3_OK;
  };
  // The two filters are independent: the front (short, memory/LDS-bound) one runs on the auxiliary stream next to the
  // background (long, f64-VALU-bound) one; the detector waits for both.
  int sm_ty = 0, sm_ntile = 0;
  size_t sm_count = 0;
  dog_pair_tiles(X, Y, &sm_ty, &sm_ntile, &sm_count);
  Scratch smaxbuf(lazy ? sm_count * (size_t)Z * sizeof(float) : 256);   // step maxima of max_im (plane-wise filter -> detector)
  if (!smaxbuf.p) return IA3_ENOMEM;
  const size_t n_strip = (lazy && Bk == 32) ? dog_pair_strips(X, Y) : 0;   // strip minima of the axis-0 result (column kernel -> bound)
  Scratch stripbuf(n_strip ? 2 * n_strip * sizeof(float) : 256);
  if (!stripbuf.p) return IA3_ENOMEM;
  float* smin_d = n_strip ? stripbuf.as<float>() : nullptr;
  float* sabs_d = n_strip ? smin_d + n_strip : nullptr;
  bool forked = false, paired = false;
  if (p.gfilt_size > 0) {
    if (p.w_front) { w.assign(p.w_front, p.w_front + 2 * p.r_front + 1); R = p.r_front; }
    else gaussian_taps(p.gfilt_size, 4.0, w, R);
    if (lazy) {
      // short stacks: both axis-0 passes from one launch (the column is loaded once), then the short filter's other two
      // axes on the auxiliary stream next to the block minima of the long filter's axis-0 result
      int fk = 0;
      rc = gauss_dog_pair(im->d, im->dtype, Z, X, Y, w.data(), R, wb.data(), Rb, a.p, b.p, tmp2.p, &fk, smaxbuf.as<float>(), smin_d, sabs_d);
      if (rc == 0) { paired = true; forked = fk != 0; }
      else if (rc != 1) { if (fk) aux_join(); return rc; }
    }
    if (!paired) {
      {
        AuxScope aux;
        forked = aux.ok;
        rc = gaussian3d(im->d, im->dtype, Z, X, Y, w.data(), R, IA3_MODE_REFLECT, a.p, tmp2.p);
      }
      if (rc) { if (forked) aux_join(); return rc; }
    }
    maxim = a.p;
  }
  if (paired) {
    minim = b.p;
    // beside the plane-wise filter (main stream): the clears and the bound, on the auxiliary stream when there is one
    std::unique_ptr<AuxResume> ar(forked ? new AuxResume() : nullptr);
    const hipStream_t bs = stream();
    rc = clear_buffers();
    if (rc) { ar.reset(); if (forked) aux_join(); return rc; }
    if (n_strip && g_strip_bound) {   // the bound from the column kernel's strip minima: no pass over the axis-0 result
      ProfScope ps("seed_blockmin");
      const int nbx = (X + 31) / 32, nby = Y / 32;
      const size_t nb = (size_t)Z * nbx * nby;
      double* lb = lazy_bound_ptr(bnd.p, nb);   // as launch_lazy lays the block out
      const size_t nt = (size_t)DOG_PAIR_ZGROUPS * nbx * nby;
      hipLaunchKernelGGL(stripbound_k, dim3((unsigned)((nt + 3) / 4)), dim3(256), 0, bs, (const float*)smin_d, (const float*)sabs_d, Z, X,
                         nbx, nby, (int)DOG_PAIR_ZGROUPS, (int)(im->dtype == IA3_U16), lb);
    } else if (im->dtype == IA3_F32) launch_lazy<float>(nullptr, b.p, Z, X, Y, wb.data(), Rb, 0, 0, bnd.p, nullptr, nullptr, nullptr, 0, nullptr, bs, 0);
    else launch_lazy<uint16_t>(nullptr, b.p, Z, X, Y, wb.data(), Rb, 0, 0, bnd.p, nullptr, nullptr, nullptr, 0, nullptr, bs, 0);
  } else if (p.background_gfilt_size > 0) {
    rc = gaussian3d(im->d, im->dtype, Z, X, Y, wb.data(), Rb, IA3_MODE_REFLECT, b.p, tmp.p, lazy ? 1 : 3);
    if (rc) { if (forked) aux_join(); return rc; }
    minim = b.p;
    if (lazy) {
      if (im->dtype == IA3_F32) launch_lazy<float>(nullptr, b.p, Z, X, Y, wb.data(), Rb, 0, 0, bnd.p, nullptr, nullptr, nullptr, 0, nullptr, s, 0);
      else launch_lazy<uint16_t>(nullptr, b.p, Z, X, Y, wb.data(), Rb, 0, 0, bnd.p, nullptr, nullptr, nullptr, 0, nullptr, s, 0);
    }
  }
  rc = clear_buffers();
  if (rc) { if (forked) aux_join(); return rc; }
  if (forked) { rc = aux_join(); if (rc) return rc; }
  const double t1 = now_ms();
  // device buffer = [SeedCtl out | SeedCtl lazy | Cand x capacity]; the header and the first FIRST candidates come back
  // in ONE copy (the common case: a few thousand seeds), the rest only if there are more
  constexpr unsigned FIRST = 8192;
  std::vector<Cand> cand;
  std::vector<char> hbuf;   // host path only (131 KB: above malloc's mmap threshold, so allocating it costs page faults)
  SeedCtl hctl, hlazy;
  for (int attempt = 0; attempt < 2; ++attempt) {
    Scratch buf(attempt == 0 ? 0 : HDR + (size_t)capacity * sizeof(Cand));
    void* bp = attempt == 0 ? buf0.p : buf.p;
    if (!bp) return IA3_ENOMEM;
    SeedCtl* dctl = (SeedCtl*)bp;
    SeedCtl* dlazy = dctl + 1;
    Cand* dcand = (Cand*)((char*)bp + HDR);
    if (attempt > 0) IA3_HIP(hipMemsetAsync(dctl, 0, HDR, s));
    if (lazy) {
      if (im->dtype == IA3_F32)
        launch_lazy<float>(maxim, b.p, Z, X, Y, wb.data(), Rb, p.min_edge_distance, th_low, bnd.p, c0buf.as<Cand0>(), dlazy, dcand, capacity, dctl, s, 1,
                           paired ? smaxbuf.as<float>() : nullptr, sm_ty, sm_ntile);
      else
        launch_lazy<uint16_t>(maxim, b.p, Z, X, Y, wb.data(), Rb, p.min_edge_distance, th_low, bnd.p, c0buf.as<Cand0>(), dlazy, dcand, capacity, dctl, s, 1,
                              paired ? smaxbuf.as<float>() : nullptr, sm_ty, sm_ntile);
    } else {
      ProfScope ps("seed_detect");
      if (im->dtype == IA3_F32)
        launch_detect<float>(p.filt_size, maxim, minim, Z, X, Y, p.min_edge_distance, th_low, dcand, capacity, dctl, s);
      else
        launch_detect<uint16_t>(p.filt_size, maxim, minim, Z, X, Y, p.min_edge_distance, th_low, dcand, capacity, dctl, s);
    }
    IA3_KCHECK();
    if (dev_finish && attempt == 0) {
      // finish on the device; the host only learns how many seeds there are
      char* fb = (char*)fin;
      hipError_t fe;
      FinCtl* fc = (FinCtl*)fb;
      const int hot_th = p.remove_hot_pixel ? p.hot_pixel_th : 0;
      void *mail_host = nullptr, *mail_dev = nullptr;
      static thread_local unsigned t_seq = 0;
      const unsigned seq = ++t_seq ? t_seq : ++t_seq;   // never 0 (the mailbox starts zeroed)
      if (host_mailbox(64, &mail_host, &mail_dev) != IA3_OK) { mail_host = mail_dev = nullptr; }
      {
        ProfScope pf("seed_finish");
        unsigned* hot = (unsigned*)(fb + o_hot);
        unsigned* rank = (unsigned*)(fb + o_rank);
        Cand* kc = (Cand*)(fb + o_kc);
        hipLaunchKernelGGL(fin_levels_k, dim3(1), dim3(1024), 0, s, (const SeedCtl*)dctl, (const Cand*)dcand, lev, p.min_dynamic_seeds, fc, kc, hot, rank);
        if (hot_th > 0)
          hipLaunchKernelGGL(fin_hot_k, dim3(FIN_KEYS / 256, FIN_S), dim3(256), 0, s, (const Cand*)kc, (const FinCtl*)fc, hot);
        hipLaunchKernelGGL(fin_rank_k, dim3(FIN_KEYS / 256, FIN_S), dim3(256), 0, s, (const Cand*)kc, fc, (const unsigned*)hot, hot_th, rank);
        hipLaunchKernelGGL(fin_scatter_k, dim3(FIN_KEYS / 256), dim3(256), 0, s, (const SeedCtl*)dctl, (const Cand*)kc, (const FinCtl*)fc,
                           (const unsigned*)hot, hot_th, (const unsigned*)rank, p.max_num_seeds, (double*)(fb + o_zxy), (double*)(fb + o_h), fc,
                           lazy ? (const SeedCtl*)dlazy : (const SeedCtl*)nullptr, LAZY_CAP, (volatile unsigned*)mail_dev, seq);
      }
      FinCtl hfc;
      fe = hipGetLastError();
      dbg_stamp("seed stage queued");
      if (fe == hipSuccess && dev->ahead) dev->ahead(dev->ahead_ctx, (const double*)(fb + o_zxy), (const unsigned*)&fc->n_alive, p.max_num_seeds);
      if (fe == hipSuccess && mail_host) {
        // the count arrives in the pinned mailbox microseconds after the kernel's store: poll it (a copy into pageable
        // memory + a sleeping synchronise cost 30-50 us of idle device between the detector and the fit)
        volatile unsigned* mb = (volatile unsigned*)mail_host;
        SpinWait sw;
        while (mb[0] != seq) {
          sw.relax();
          if (((sw.n & 0xfffff) == 0 || (sw.n > 40400 && (sw.n & 0x3ff) == 0)) && hipStreamQuery(s) != hipErrorNotReady) {   // the stream drained (or failed) without the word
            if (mb[0] == seq) break;
            fe = hipStreamSynchronize(s);
            if (fe == hipSuccess && mb[0] != seq) fe = hipErrorUnknown;
            break;
          }
        }
        hfc.n_alive = mb[1]; hfc.chosen = (int)mb[2]; hfc.n_cand = mb[3]; hfc.overflow = mb[4];
        dbg_stamp("seed count seen");
      } else {
        if (fe == hipSuccess) fe = hipMemcpyAsync(&hfc, fc, sizeof(FinCtl), hipMemcpyDeviceToHost, s);
        if (fe == hipSuccess) fe = hipStreamSynchronize(s);
      }
      if (fe != hipSuccess) return set_error(IA3_EHIP, "seed finish failed: %s", hipGetErrorString(fe));
      if (hfc.overflow & 2u)     // more first-stage candidates than the lazy path is sized for: dense filter instead
        return dog_seed_impl(im, p, out, dev, true);
      hctl.n_cand = hfc.n_cand; hctl.overflow = hfc.overflow;
      if (hctl.n_cand <= FIN_CAP && !hctl.overflow) {
        int n = (int)hfc.n_alive;
        if (p.max_num_seeds > 0 && p.max_num_seeds <= n) n = p.max_num_seeds;
        dev->on_device = true;
        dev->n = n;
        dev->th_used = lev.th[hfc.chosen];
        dev->d_zxy = (const double*)(fb + o_zxy);
        dev->d_h = (const double*)(fb + o_h);
        dev->hold = fin;
        fin = nullptr;   // now the caller's
        if (dbg) fprintf(stderr, "dog_seed: device finish, %u candidates -> %d seeds\n", hctl.n_cand, n);
        return IA3_OK;
      }
      ws_put(fin);   // too many candidates: the host path below takes over (the detector's output is still in buf)
      fin = nullptr;
    }
    hbuf.resize(HDR + (size_t)FIRST * sizeof(Cand));
    IA3_HIP(hipMemcpyAsync(hbuf.data(), bp, hbuf.size(), hipMemcpyDeviceToHost, s));
    IA3_HIP(hipStreamSynchronize(s));
    memcpy(&hctl, hbuf.data(), sizeof(SeedCtl));
    memcpy(&hlazy, hbuf.data() + sizeof(SeedCtl), sizeof(SeedCtl));
    if (lazy && (hlazy.overflow || hlazy.n_cand > LAZY_CAP)) return dog_seed_impl(im, p, out, dev, true);
    if (hctl.n_cand <= capacity) {
      cand.resize(hctl.n_cand);
      const unsigned nfirst = hctl.n_cand < FIRST ? hctl.n_cand : FIRST;
      if (nfirst) memcpy(cand.data(), hbuf.data() + HDR, (size_t)nfirst * sizeof(Cand));
      if (hctl.n_cand > FIRST) {
        IA3_HIP(hipMemcpyAsync(cand.data() + FIRST, dcand + FIRST, (size_t)(hctl.n_cand - FIRST) * sizeof(Cand), hipMemcpyDeviceToHost, s));
        IA3_HIP(hipStreamSynchronize(s));
      }
      break;
    }
    if (attempt == 1) return set_error(IA3_ECAPACITY, "more than %u seed candidates", capacity);
    capacity = hctl.n_cand + 1024;  // exact size known now; one retry
  }
  const double t2 = now_ms();
  finish_seeds(cand, lev, p, Y, out);
  if (dbg) fprintf(stderr, "dog_seed: launch gauss %.3f ms, detect+copy(sync) %.3f ms, finish %.3f ms (%zu cand)\n", t1 - t0, t2 - t1, now_ms() - t2, cand.size());
  return IA3_OK;
}

int dog_seed_dev(const ia3_stack* im, const ia3_seed_params& p, SeedDev& out) {
  return dog_seed_impl(im, p, out.host, &out);
}

int dog_seed(const ia3_stack* im, const ia3_seed_params& p, SeedOut& out) {
  SeedDev d;
  int rc = dog_seed_impl(im, p, d.host, &d); if (rc) return rc;
  if (!d.on_device) { out = std::move(d.host); return IA3_OK; }
  out.th_used = d.th_used;
  out.zxyh.assign((size_t)d.n * 4, 0.0);
  if (d.n) {
    std::vector<double> zxy((size_t)d.n * 3), hh((size_t)d.n);
    hipStream_t s = stream();
    IA3_HIP(hipMemcpyAsync(zxy.data(), d.d_zxy, zxy.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    IA3_HIP(hipMemcpyAsync(hh.data(), d.d_h, hh.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    IA3_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < d.n; ++i) {
      out.zxyh[4 * i] = zxy[3 * i]; out.zxyh[4 * i + 1] = zxy[3 * i + 1]; out.zxyh[4 * i + 2] = zxy[3 * i + 2];
      out.zxyh[4 * i + 3] = hh[i];
    }
  }
  return IA3_OK;
}

// ---- legacy seeding (visual_tools.py:348-381 get_seed_points_base, :1775-1870 get_seed_in_distance) ----------
// Filters + rule-1 detection on a resident stack; candidates come back in np.where order (z, x, y ascending).
static int legacy_candidates(const ia3_stack* im, double gfilt, double bgfilt, int filt_size, double th_low,
                             std::vector<Cand>& cand) {
  hipStream_t s = stream();
  const int Z = im->Z, X = im->X, Y = im->Y;
  const size_t bytes = im->bytes;
  if (filt_size < 1 || filt_size > 8) return set_error(IA3_EUNSUPPORTED, "filt_size %d not in 1..8", filt_size);
  Scratch a(bytes), b(bytes), tmp(bytes);
  if (!a.p || !b.p || !tmp.p) return IA3_ENOMEM;
  const void* maxim = im->d;
  const void* minim = im->d;
  std::vector<double> w; int R, rc;
  if (gfilt > 0) {     // `if gfilt_size:` — scipy defaults: mode='reflect', truncate=4
    gaussian_taps(gfilt, 4.0, w, R);
    rc = gaussian3d(im->d, im->dtype, Z, X, Y, w.data(), R, IA3_MODE_REFLECT, a.p, tmp.p);
    if (rc) return rc;
    maxim = a.p;
  }
  if (bgfilt > 0) {
    gaussian_taps(bgfilt, 4.0, w, R);
    rc = gaussian3d(im->d, im->dtype, Z, X, Y, w.data(), R, IA3_MODE_REFLECT, b.p, tmp.p);
    if (rc) return rc;
    minim = b.p;
  }
  unsigned capacity = 1u << 16;
  for (int attempt = 0; attempt < 2; ++attempt) {
    Scratch buf(sizeof(SeedCtl) + (size_t)capacity * sizeof(Cand));
    if (!buf.p) return IA3_ENOMEM;
    SeedCtl* dctl = (SeedCtl*)buf.p;
    Cand* dcand = (Cand*)((char*)buf.p + sizeof(SeedCtl));
    IA3_HIP(hipMemsetAsync(dctl, 0, sizeof(SeedCtl), s));
    if (im->dtype == IA3_F32)
      launch_detect<float>(filt_size, maxim, minim, Z, X, Y, 0, th_low, dcand, capacity, dctl, s, 1);
    else
      launch_detect<uint16_t>(filt_size, maxim, minim, Z, X, Y, 0, th_low, dcand, capacity, dctl, s, 1);
    IA3_KCHECK();
    SeedCtl hctl;
    IA3_HIP(hipMemcpyAsync(&hctl, dctl, sizeof(SeedCtl), hipMemcpyDeviceToHost, s));
    IA3_HIP(hipStreamSynchronize(s));
    if (hctl.n_cand <= capacity) {
      cand.resize(hctl.n_cand);
      if (hctl.n_cand) {
        IA3_HIP(hipMemcpyAsync(cand.data(), dcand, (size_t)hctl.n_cand * sizeof(Cand), hipMemcpyDeviceToHost, s));
        IA3_HIP(hipStreamSynchronize(s));
      }
      break;
    }
    if (attempt == 1) return set_error(IA3_ECAPACITY, "more than %u seed candidates", capacity);
    capacity = hctl.n_cand + 1024;
  }
  std::sort(cand.begin(), cand.end(), [](const Cand& p, const Cand& q) {
    if (p.z != q.z) return p.z < q.z;
    if (p.x != q.x) return p.x < q.x;
    return p.y < q.y;
  });
  return IA3_OK;
}

// get_seed_points_base's tail for one threshold: keep h > th, drop (x, y) columns seen more than hot_pix_th times
static void legacy_base(const std::vector<Cand>& cand, double th, int hot_pix_th, std::vector<Cand>& out) {
  out.clear();
  for (auto& k : cand) if ((double)k.h > th) out.push_back(k);
  if (hot_pix_th > 0 && !out.empty()) {
    std::vector<std::pair<long long, int>> keys;
    keys.reserve(out.size());
    for (auto& k : out) keys.push_back({((long long)k.x << 32) | (unsigned)k.y, 0});
    std::vector<long long> sorted;
    sorted.reserve(keys.size());
    for (auto& k : keys) sorted.push_back(k.first);
    std::sort(sorted.begin(), sorted.end());
    std::vector<Cand> kept;
    for (size_t i = 0; i < out.size(); ++i) {
      auto r = std::equal_range(sorted.begin(), sorted.end(), keys[i].first);
      if ((r.second - r.first) <= hot_pix_th) kept.push_back(out[i]);
    }
    out.swap(kept);
  }
}

int seed_in_distance(const void* im, int dtype, int Z, int X, int Y, const double* center,
                     const ia3_legacy_seed_params& p, std::vector<long long>& zxyh) {
  const size_t es = dtype == IA3_F32 ? 4 : 2;
  int lo[3] = {0, 0, 0}, hi[3] = {Z, X, Y};
  double lc[3] = {0, 0, 0};
  if (center) {   // visual_tools.py:1817-1830: half radius along z, truncation to int
    const int dim[3] = {Z, X, Y};
    for (int a = 0; a < 3; ++a) {
      const double r = a == 0 ? p.seed_radius / 2 : p.seed_radius;
      double l = center[a] - r; if (l < 0) l = 0;
      double h = center[a] + r; if (h > dim[a]) h = dim[a];
      lo[a] = (int)l; hi[a] = (int)h;
      lc[a] = center[a] - lo[a];
    }
    if (hi[0] <= lo[0] || hi[1] <= lo[1] || hi[2] <= lo[2]) { zxyh.clear(); return IA3_OK; }
  }
  const int cz = hi[0] - lo[0], cx = hi[1] - lo[1], cy = hi[2] - lo[2];
  std::vector<char> crop((size_t)cz * cx * cy * es);
  for (int z = 0; z < cz; ++z)
    for (int x = 0; x < cx; ++x)
      memcpy(crop.data() + ((size_t)z * cx + x) * cy * es,
             (const char*)im + ((((size_t)(z + lo[0])) * X + (x + lo[1])) * Y + lo[2]) * es, (size_t)cy * es);
  ia3_stack* st = nullptr;
  int rc = ia3_stack_upload(crop.data(), dtype, cz, cx, cy, &st);
  if (rc) return rc;
  std::vector<Cand> cand, sel, seeds;
  const bool dyn = center && p.dynamic;
  // the non-dynamic calls (:1851-1858) do not forward background_gfilt_size: the base default (10) applies
  const double bg = dyn ? p.background_gfilt_size : 10.0;
  std::vector<double> levels;
  if (dyn) {   // np.linspace(1, 1/iters, iters) * th_seed
    const int n = p.dynamic_iters;
    if (n < 1) { ia3_stack_free(st); return set_error(IA3_EINVAL, "dynamic_iters must be >= 1"); }
    const double start = 1.0, stop = 1.0 / n;
    const double div = n > 1 ? n - 1 : 1, delta = stop - start, step = delta / div;
    for (int i = 0; i < n; ++i) {
      double y = step != 0 ? i * step : (i / div) * delta;
      levels.push_back(y + start);
    }
    if (n > 1) levels[n - 1] = stop;
    for (auto& l : levels) l = p.th_seed * l;
  } else {
    levels.push_back(p.th_seed);
  }
  double th_low = levels[0];
  for (double l : levels) th_low = l < th_low ? l : th_low;
  rc = legacy_candidates(st, p.gfilt_size, bg, p.filt_size, th_low, cand);
  ia3_stack_free(st);
  if (rc) return rc;
  if (dyn) {
    for (double th : levels) {
      legacy_base(cand, th, p.hot_pix_th, sel);
      seeds.clear();
      for (auto& k : sel) {
        const double dz = k.z - lc[0], dx = k.x - lc[1], dy = k.y - lc[2];
        if (sqrt(dz * dz + dx * dx + dy * dy) < p.seed_radius) {
          Cand c = k; c.z += lo[0]; c.x += lo[1]; c.y += lo[2];
          seeds.push_back(c);
        }
      }
      const int n = (int)seeds.size();
      if (p.num_seeds > 0 && n >= (p.num_seeds < p.min_dynamic_seeds ? p.num_seeds : p.min_dynamic_seeds)) break;
      if (p.num_seeds == 0 && n >= p.min_dynamic_seeds) break;
    }
  } else {
    // :1851-1858: crop coordinates are returned as they are (no offset, no distance test) in these branches
    legacy_base(cand, p.th_seed, p.hot_pix_th, seeds);
  }
  if (seeds.size() > 1) {
    // np.argsort(h) ascending, last num_seeds, flipped.  Equal heights: stable order assumed (NumPy's default
    // sort leaves it implementation-defined)
    std::stable_sort(seeds.begin(), seeds.end(), [](const Cand& a, const Cand& b) { return a.h < b.h; });
    if (p.num_seeds > 0 && (size_t)p.num_seeds < seeds.size()) seeds.erase(seeds.begin(), seeds.end() - p.num_seeds);
    std::reverse(seeds.begin(), seeds.end());
  }
  zxyh.resize(seeds.size() * 4);
  for (size_t i = 0; i < seeds.size(); ++i) {
    zxyh[4 * i] = seeds[i].z; zxyh[4 * i + 1] = seeds[i].x; zxyh[4 * i + 2] = seeds[i].y; zxyh[4 * i + 3] = (long long)seeds[i].h;
  }
  return IA3_OK;
}

}  // namespace ia3k

extern "C" {

int ia3_seed_in_distance(const void* im, int dtype, int Z, int X, int Y, const double* center,
                         const ia3_legacy_seed_params* p, int64_t* out_zxyh, int capacity, int* n_out) {
  int rc = ensure_init(); if (rc) return rc;
  if (!im || !p || !n_out) return set_error(IA3_EINVAL, "null argument");
  if (dtype != IA3_F32 && dtype != IA3_U16) return set_error(IA3_EINVAL, "dtype must be IA3_U16 or IA3_F32");
  std::vector<long long> o;
  rc = ia3k::seed_in_distance(im, dtype, Z, X, Y, center, *p, o); if (rc) return rc;
  const int n = (int)(o.size() / 4);
  *n_out = n;
  if (n > capacity) return set_error(IA3_ECAPACITY, "seed buffer too small: need %d rows", n);
  if (n && !out_zxyh) return set_error(IA3_EINVAL, "null output");
  for (size_t i = 0; i < o.size(); ++i) out_zxyh[i] = (int64_t)o[i];
  return IA3_OK;
}


int ia3_dog_seed_dev(const ia3_stack* im, const ia3_seed_params* p, double* out_zxyh, int capacity,
                     int* n_out, double* th_used) {
  int rc = ensure_init(); if (rc) return rc;
  if (!im || !p || !n_out) return set_error(IA3_EINVAL, "null argument");
  ia3k::SeedOut o;
  rc = ia3k::dog_seed(im, *p, o); if (rc) return rc;
  int n = (int)(o.zxyh.size() / 4);
  *n_out = n;
  if (th_used) *th_used = o.th_used;
  if (n > capacity) return set_error(IA3_ECAPACITY, "seed buffer too small: need %d rows", n);
  if (n && !out_zxyh) return set_error(IA3_EINVAL, "null output");
  if (n) memcpy(out_zxyh, o.zxyh.data(), o.zxyh.size() * sizeof(double));
  return IA3_OK;
}

int ia3_dog_filters_dev(const ia3_stack* im, double sigma_front, double sigma_back, ia3_stack* front, ia3_stack* back_axis0) {
  int rc = ensure_init(); if (rc) return rc;
  if (!im || !front || !back_axis0) return set_error(IA3_EINVAL, "null stack");
  for (const ia3_stack* o : {front, back_axis0})
    if (o->dtype != im->dtype || o->Z != im->Z || o->X != im->X || o->Y != im->Y) return set_error(IA3_EINVAL, "stacks differ in shape or dtype");
  if (front->d == im->d || back_axis0->d == im->d || front->d == back_axis0->d) return set_error(IA3_EINVAL, "outputs must not alias");
  if (!(sigma_front > 0) || !(sigma_back > 0)) return set_error(IA3_EINVAL, "sigma must be > 0");
  std::vector<double> wf, wb; int rf, rb;
  gaussian_taps(sigma_front, 4.0, wf, rf);
  gaussian_taps(sigma_back, 4.0, wb, rb);
  Scratch tmp(im->bytes);
  if (!tmp.p) return IA3_ENOMEM;
  int forked = 0;
  rc = ia3k::gauss_dog_pair(im->d, im->dtype, im->Z, im->X, im->Y, wf.data(), rf, wb.data(), rb, front->d, back_axis0->d, tmp.p, &forked);
  if (rc == 0) return forked ? aux_join() : IA3_OK;
  if (rc != 1) { if (forked) aux_join(); return rc; }
  rc = ia3k::gaussian3d(im->d, im->dtype, im->Z, im->X, im->Y, wf.data(), rf, IA3_MODE_REFLECT, front->d, tmp.p);
  if (rc) return rc;
  return ia3k::gaussian3d(im->d, im->dtype, im->Z, im->X, im->Y, wb.data(), rb, IA3_MODE_REFLECT, back_axis0->d, tmp.p, 1);
}

int ia3_dog_seed(const void* im, int dtype, int Z, int X, int Y, const ia3_seed_params* p,
                 double* out_zxyh, int capacity, int* n_out, double* th_used) {
  ia3_stack* a = nullptr;
  int rc = ia3_stack_upload(im, dtype, Z, X, Y, &a); if (rc) return rc;
  rc = ia3_dog_seed_dev(a, p, out_zxyh, capacity, n_out, th_used);
  ia3_stack_free(a);
  return rc;
}

}  // extern "C"
