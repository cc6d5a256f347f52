// Separable 3-D Gaussian with scipy.ndimage.gaussian_filter's exact arithmetic, for gfx950.
//
// Contract (verified bit-for-bit against SciPy 1.15 by tests):
//   per axis 0,1,2:  out[i] = in[i]*w0 + sum_{j=R..1} (in[i-j] + in[i+j]) * w_j
//   accumulated in float64 in exactly that order (NI_Correlate1D's symmetric branch), no FMA
//   contraction (this file is compiled with -ffp-contract=off), result stored to the stack dtype
//   after every axis (float32: round-to-nearest; uint16: truncation).
//
// Roofline note: a 61-tap pass costs 91 f64 VALU ops per voxel against
// 8 B of HBM traffic, i.e. 11 flop/B where the machine balance for non-FMA f64 is ~6 flop/B, so
// these kernels are f64-VALU-bound, not HBM-bound.  They are therefore organised around the
// VALU: every thread keeps a window of K+2R inputs in registers (converted to f64 once) and
// produces K consecutive outputs along the filter axis, so the inner loop is three VALU
// instructions per tap pair with the taps in SGPRs and no LDS/VMEM traffic.
//   - axes 0/1 (strided): lanes run along the contiguous y axis, every window load is a coalesced
//     256 B row segment per wave; the reflect/nearest index mapping is wave-uniform (scalar).
//   - axis 2 (contiguous), short filters (HBM-bound): a wave owns 64 rows; 64x32 tiles are transposed through
//     LDS so that every lane slides the same register window along its own row while global access stays
//     row-coalesced.  Long filters (VALU-bound, R >= 16): the axis-1 pass stores each plane transposed and the
//     axis-2 pass is the strided kernel over that transposed plane, storing transposed again.
#include "ia3_gauss.h"
#include <unistd.h>
#include <cstring>
#include <mutex>

using namespace ia3g;

namespace {

// bmap[i] = border-mapped source index of position (i - R), i in [0, count); positions past the
// end of the last chunk are only ever loaded, never stored, and map to valid indices as well.
__global__ void border_map_k(int* bmap, int count, int R, int len, int mode) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < count) bmap[i] = border_idx(i - R, len, mode);
}  // w[j] = weight at offset j (0..R), passed by value -> SGPRs

// ---- strided axis: element(line p, position q) = base + q*stride + p ---------------------------
// A thread walks one segment [blockIdx.z*seg, +seg) of its line in chunks of K outputs with a sliding register
// window: after a chunk the window shifts by K (2R register moves) and only K new inputs are loaded, so every
// input is read once per segment instead of (K+2R)/K times.
// TR: the output is written TRANSPOSED inside each outer slab, out[outer][p][q] instead of out[outer][q][p].  Each
// wave parks four chunks (32 consecutive q of its 64 lines) in a private 64 x 32 LDS tile and writes it out as
// 128-B row pieces, two rows per store instruction, so the transposed store stays coalesced.  Two transposing
// passes in a row (axis 1, then axis 2 run as a strided pass over the transposed slab) give the contiguous axis
// the same register-window kernel as the other two.
template <class T, int R, int K, bool TR = false>
__global__ __launch_bounds__(256) void gauss_strided(const T* __restrict__ in, T* __restrict__ out,
                                                     int inner, size_t stride, int len,
                                                     size_t outer_stride, Taps taps,
                                                     const int* __restrict__ bmap, int seg, int cert) {
  constexpr bool FAST = R >= 16;   // certified fused path only where the pass is VALU-bound
  constexpr int TQ = (K == 6 || K == 12) ? 24 : ((K == 10) ? 30 : 32);   // q extent of the transposing tile (a multiple of K)
  static_assert(!TR || TQ % K == 0, "tile must hold whole chunks");
  __shared__ float tile[TR ? 4 : 1][TR ? 64 : 1][TR ? TQ + 1 : 1];
  const int p_raw = blockIdx.x * 256 + threadIdx.x;
  if (!TR && p_raw >= inner) return;
  const int p = p_raw < inner ? p_raw : inner - 1;   // TR: lanes past the end load a valid line and store nothing
  const size_t base = (size_t)blockIdx.y * outer_stride + p;
  const int q_begin = blockIdx.z * seg;
  const int q_end = q_begin + seg < len ? q_begin + seg : len;
  // bmap[i] = border-mapped index of position i - R (wave-uniform -> scalar loads)
  // The window is a ring of W = K + 2R registers.  When K divides W the chunk loop is unrolled W/K times and the
  // ring is indexed with compile-time offsets, so sliding costs nothing; otherwise (ROT = 1) the window is shifted
  // with 2R register moves per chunk.
  constexpr int W0 = K + 2 * R;                                    // samples a chunk needs
  constexpr bool RING = (W0 % K == 0) || FAST;                     // long filters: pad the ring to a multiple of K
  constexpr int W = RING ? ((W0 + K - 1) / K) * K : W0;            // (spare slots take the prefetched inputs)
  constexpr int U = RING ? W / K : 1;   // chunks per unrolled body
  double win[W];
  unsigned sbits = 0;   // OR of the raw inputs this thread has loaded: bit 31 set = a negative (or -0, nan) was seen
#pragma unroll
  for (int i = 0; i < W0; ++i) {
    const T v = in[base + (size_t)bmap[q_begin + i] * stride];
    if constexpr (FAST) sbits |= sign_of<T>(v);
    win[i] = (double)v;
  }
#pragma unroll
  for (int i = W0; i < W; ++i) win[i] = 0.0;
  for (int qq = q_begin; qq < q_end; qq += K * U) {
    // the U chunks of one turn of the ring, expanded at compile time (static_for_until: stops at the segment end)
    auto chunk = [&](auto cc) -> bool {
      constexpr int c = decltype(cc)::value;
      const int q0 = qq + c * K;
      if (q0 >= q_end) return false;            // wave-uniform
      constexpr int o = (U == 1) ? 0 : c * K;   // physical slot of logical window index 0
      // prefetch the K inputs the next chunk adds (positions q0+K+R .. q0+2K+R-1)
      T nxt[K];   // kept in the stack dtype (half the registers of f64) until they enter the window
#pragma unroll
      for (int i = 0; i < K; ++i) nxt[i] = in[base + (size_t)bmap[q0 + K + 2 * R + i] * stride];
      double acc[K];
      bool redo = true;
      if constexpr (FAST) {
        if (cert >= 0 && (int)sbits >= 0) {
#pragma unroll
          for (int k = 0; k < K; ++k) acc[k] = win[(o + k + R) % W] * taps.w[0];
#pragma unroll
          for (int j = R; j >= 1; --j) {
#pragma unroll
            for (int k = 0; k < K; ++k)
              acc[k] = __builtin_fma(win[(o + k + R - j) % W] + win[(o + k + R + j) % W], taps.w[j], acc[k]);
          }
          redo = false;
#pragma unroll
          for (int k = 0; k < K; ++k) redo |= uncertain<T>(acc[k], cert);
        }
      }
      if (redo) {   // the reference sequence: separate multiply and add, same order
        // `one` is 1.0 the compiler cannot see through: without it the pair sums of this (rare) branch are shared
        // with the fused branch above and all 30*K of them are kept live across the branch (x * 1.0 is exact)
        double one = 1.0;
        if constexpr (FAST) asm volatile("" : "+v"(one));
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = win[(o + k + R) % W] * taps.w[0];
#pragma unroll
        for (int j = R; j >= 1; --j) {
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const double a = FAST ? win[(o + k + R - j) % W] * one : win[(o + k + R - j) % W];
            acc[k] = acc[k] + (a + win[(o + k + R + j) % W]) * taps.w[j];
          }
        }
      }
      if constexpr (FAST) {
#pragma unroll
        for (int i = 0; i < K; ++i) sbits |= sign_of<T>(nxt[i]);
      }
      if constexpr (TR) {
        const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
        const int cq = ((q0 - q_begin) / K) % (TQ / K);     // chunk slot inside the tile (wave-uniform)
#pragma unroll
        for (int k = 0; k < K; ++k) tile[wv][ln][cq * K + k] = (float)cvt<T>(acc[k]);   // exact in a float
        if (cq == TQ / K - 1 || q0 + K >= q_end) {
          __builtin_amdgcn_wave_barrier();
          const int qt0 = q0 - cq * K;
          const int nq = q_end - qt0 < TQ ? q_end - qt0 : TQ;
          const int col = ln & 31, hrow = ln >> 5;
          const int pw = blockIdx.x * 256 + wv * 64;
          T* op = out + (size_t)blockIdx.y * outer_stride + qt0 + col;
#pragma unroll 8
          for (int r2 = 0; r2 < 32; ++r2) {
            const int row = 2 * r2 + hrow;
            if (col < nq && pw + row < inner) op[(size_t)(pw + row) * len] = (T)tile[wv][row][col];
          }
          __builtin_amdgcn_wave_barrier();
        }
      } else {
#pragma unroll
        for (int k = 0; k < K; ++k)
          if (q0 + k < q_end) out[base + (size_t)(q0 + k) * stride] = cvt<T>(acc[k]);
      }
      if constexpr (U == 1) {
#pragma unroll
        for (int i = 0; i < 2 * R; ++i) win[i] = win[i + K];
#pragma unroll
        for (int i = 0; i < K; ++i) win[2 * R + i] = (double)nxt[i];
      } else {
#pragma unroll
        for (int i = 0; i < K; ++i) win[(o + W0 + i) % W] = (double)nxt[i];   // spare slots first, then the oldest ones
      }
      return true;
    };
    static_for_until<0, U>(chunk);
  }
}

// ---- contiguous axis: a wave owns 64 rows and walks a y-segment with the same sliding register window ------
// Lanes must run along y for coalesced global access but a thread wants its own row, so the data goes through
// LDS transposes in 64 (rows) x 32 (y) tiles: two rows per wave-load fill tin[row][y] with 128-B row pieces; lane r
// then reads ITS row (stride 33 words -> conflict-free), slides the window over the 32 new inputs in chunks of K,
// drops the outputs into tout[row][y], and the tile is written back the same way.  Every input is read once.
template <class T, int R, int K>
__global__ __launch_bounds__(64, 2) void gauss_contig(const T* __restrict__ in, T* __restrict__ out, int len,
                                                   size_t n_rows, Taps taps, const int* __restrict__ bmap, int seg) {
  constexpr int TW = 32;
  static_assert(TW % K == 0, "a tile must hold whole chunks");
  __shared__ float tin[64][TW + 1];
  __shared__ float tout[64][TW + 1];
  const int lane = threadIdx.x, half = lane >> 5, col = lane & 31;
  const size_t row0 = (size_t)blockIdx.x * 64;
  const int nr = n_rows - row0 < 64 ? (int)(n_rows - row0) : 64;   // rows of this block
  const int q_begin = blockIdx.y * seg;
  const int q_end = q_begin + seg < len ? q_begin + seg : len;
  // bmap[i] = border-mapped index of position i - R
  double win[K + 2 * R];
  // prologue: positions q_begin-R .. q_begin+R-1 (2R values per row), streamed through tin in pieces of TW
#pragma unroll
  for (int p0 = 0; p0 < 2 * R; p0 += TW) {
    const int w = 2 * R - p0 < TW ? 2 * R - p0 : TW;
    __syncthreads();
    for (int r = half; r < nr; r += 2)
      if (col < w) tin[r][col] = (float)ld<T>(in, (row0 + r) * (size_t)len + bmap[q_begin + p0 + col]);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < TW; ++c)
      if (c < w) win[p0 + c] = (double)tin[lane][c];
  }
  // tile pipeline: the inputs of tile t+1 travel HBM -> registers while tile t is being filtered out of LDS
  T pre[32];
  auto fetch = [&](int o0) {   // inputs at positions o0+R .. o0+R+TW-1  ->  bmap index = position + R
    const int src = bmap[o0 + 2 * R + col];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int r = half + 2 * i;
      pre[i] = r < nr ? in[(row0 + r) * (size_t)len + src] : (T)0;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < 32; ++i) tin[half + 2 * i][col] = (float)pre[i];
  };
  __syncthreads();
  fetch(q_begin);
  stash();
  __syncthreads();
  for (int o0 = q_begin; o0 < q_end; o0 += TW) {
    const bool more = o0 + TW < q_end;
    if (more) fetch(o0 + TW);
#pragma unroll 1
    for (int c8 = 0; c8 < TW; c8 += K) {   // not unrolled: one chunk's registers at a time
#pragma unroll
      for (int i = 0; i < K; ++i) win[2 * R + i] = (double)tin[lane][c8 + i];
      double acc[K];
#pragma unroll
      for (int k = 0; k < K; ++k) acc[k] = win[k + R] * taps.w[0];
#pragma unroll
      for (int j = R; j >= 1; --j) {
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = acc[k] + (win[k + R - j] + win[k + R + j]) * taps.w[j];
      }
      // quantise to the stack dtype here (float32 round / uint16 truncate); the value is exact in a float
#pragma unroll
      for (int k = 0; k < K; ++k) tout[lane][c8 + k] = (float)cvt<T>(acc[k]);
#pragma unroll
      for (int i = 0; i < 2 * R; ++i) win[i] = win[i + K];
    }
    __syncthreads();   // every lane is done with tin and has filled its row of tout
    const int wv = q_end - o0 < TW ? q_end - o0 : TW;
#pragma unroll 4
    for (int r = half; r < nr; r += 2)
      if (col < wv) out[(row0 + r) * (size_t)len + o0 + col] = (T)tout[r][col];
    if (more) stash();
    __syncthreads();
  }
}

// ---- generic fallback: any radius, one output per thread, taps from global memory --------------
template <class T>
__global__ __launch_bounds__(256) void gauss_generic(const T* __restrict__ in, T* __restrict__ out,
                                                     int inner, size_t stride, int len, size_t outer_stride,
                                                     size_t inner_stride, const double* __restrict__ w,
                                                     int R, int mode) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= inner) return;
  const size_t base = (size_t)blockIdx.y * outer_stride + (size_t)p * inner_stride;
  for (int q = blockIdx.z; q < len; q += gridDim.z) {
    double acc = ld<T>(in, base + (size_t)q * stride) * w[0];
    for (int j = R; j >= 1; --j) {
      double a = ld<T>(in, base + (size_t)border_idx(q - j, len, mode) * stride);
      double b = ld<T>(in, base + (size_t)border_idx(q + j, len, mode) * stride);
      acc = acc + (a + b) * w[j];
    }
    out[base + (size_t)q * stride] = cvt<T>(acc);
  }
}

template <class T>
__global__ __launch_bounds__(256) void highpass_k(const T* __restrict__ im, const T* __restrict__ low,
                                                  T* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t step = (size_t)gridDim.x * 256;
  for (; i < n; i += step) {
    T a = im[i], b = low[i];
    out[i] = b > a ? (T)0 : (T)(a - b);
  }
}

// ---- short filters, all three axes in one kernel --------------------------------------------------------------
// The three passes of a short filter are HBM-bound (10 f64 ops per 8 B for R = 3), so running them back to back
// costs three reads and three writes of the stack.  Here a 256-thread block owns a TX x TY (x, y) tile and marches
// along z: the (TX+2R) x (TY+2R) halo tile of the axis-0 result of plane z is produced from a (2R+1)-deep register
// window of raw planes (each thread owns ~11 positions of the halo tile), quantised to the stack dtype exactly as
// the standalone pass would store it, and parked in LDS; the axis-1 pass reads it out of LDS (TX x (TY+2R)
// outputs, quantised, second LDS tile) and the axis-2 pass reads that and stores to HBM with lanes along y.  Every
// intermediate value is the one the three-kernel path produces (same order of operations, same rounding to the
// stack dtype between axes); traffic drops from 6 to ~2.3 stack transfers.
// Workgroups are handed to the 8 XCDs round-robin by linear id, and each XCD has its own L2: tiles that share halo
// lines must land on the same XCD to share them.  Map linear id b to the tile (b % 8) * ceil(n/8) + b / 8, so every
// XCD walks a contiguous run of tiles (neighbours in y are 8 dispatch slots apart, resident together).  Returns -1
// for the padding ids of the last run.
__device__ __forceinline__ int xcd_tile(int b, int n_tiles) {
  const int per = (n_tiles + 7) >> 3;
  const int t = (b & 7) * per + (b >> 3);
  return ((b >> 3) < per && t < n_tiles) ? t : -1;
}

template <int R> constexpr int fused_threads() { return R <= 3 ? 512 : 256; }

template <class T, int R>
__global__ __launch_bounds__(fused_threads<R>()) void gauss3_fused(const T* __restrict__ in, T* __restrict__ out, int Z, int X, int Y,
                                                    Taps taps, const int* __restrict__ mz, const int* __restrict__ mx,
                                                    const int* __restrict__ my, int zseg) {
  constexpr int TX = 32, TY = 64, W = 2 * R + 1, NT = fused_threads<R>();
  constexpr int EX = TX + 2 * R, EY = TY + 2 * R;          // halo tile
  constexpr int NPOS = EX * EY, SL = (NPOS + NT - 1) / NT;   // positions per thread
  constexpr int NX1 = TX * EY, S1 = (NX1 + NT - 1) / NT;     // axis-1 outputs per thread
  constexpr int S2 = TX * TY / NT;                        // axis-2 outputs per thread
  // SQ counters put the first version of this kernel at 78 VALU instructions per voxel for 30 of arithmetic: seven
  // f32->f64 conversions per tap set on each of the three axes, six register moves per position to slide the z
  // window, index arithmetic.  For R <= 3 the window therefore holds float64 values (one conversion per loaded voxel)
  // in a ring addressed with compile-time offsets (the z loop is expanded W times, no moves), with 512 threads per
  // block so that the 2 x W x SL window registers fit.  Alone the kernel is not faster (0.91 vs 0.81 ms, fewer blocks
  // in flight), but it runs side by side with the VALU-bound first long pass (seed.hip), and there the VALU
  // instructions it no longer issues are what counts: DoG stage 3.26 -> 3.06 ms.  Float64 LDS planes (no conversion
  // on the 2 x W reads per output) and 1024-thread blocks were measured as well: 3.12 and 3.23-3.33 ms.  Larger radii
  // keep the narrow window: W x SL float64 values would not fit the register file.
  constexpr bool WIDE = R <= 3;
  using WT = std::conditional_t<WIDE, double, T>;          // window element
  using LT = float;                                        // LDS element (the quantised intermediate is exact in a float)
  constexpr int U = WIDE ? W : 1;                          // z steps per expanded loop body
  constexpr int PADY = WIDE ? 2 : 1;                       // WIDE: rows of 72 floats, 16-byte aligned for the axis-2 pieces
  __shared__ __attribute__((aligned(16))) LT A[EX][EY + PADY];
  __shared__ __attribute__((aligned(16))) LT B[TX][EY + PADY];
  const int tid = threadIdx.x;
  const int nty = (Y + TY - 1) / TY, ntx = (X + TX - 1) / TX;
  const int tile_id = xcd_tile(blockIdx.x, nty * ntx);   // grid.x = 8 * ceil(tiles / 8)
  if (tile_id < 0) return;                               // whole block leaves together
  const int x0 = (tile_id / nty) * TX, y0 = (tile_id % nty) * TY;
  const int z_begin = blockIdx.z * zseg;
  const int z_end = z_begin + zseg < Z ? z_begin + zseg : Z;
  const size_t plane = (size_t)X * Y;
  // in-plane offsets of this thread's halo positions (border-mapped: mx/my[i] = source index of position i - R)
  int off[SL];
#pragma unroll
  for (int k = 0; k < SL; ++k) {
    const int pos = tid + NT * k;
    const int xx = pos / EY, yy = pos - xx * EY;
    off[k] = pos < NPOS ? mx[x0 + xx] * Y + my[y0 + yy] : 0;
  }
  WT win[SL][W];   // planes z-R .. z+R of every owned position; logical slot j lives at (phase + j) % W when WIDE
#pragma unroll
  for (int j = 0; j < W - 1; ++j) {
    const size_t pz = (size_t)mz[z_begin + j] * plane;
#pragma unroll
    for (int k = 0; k < SL; ++k) win[k][j] = (WT)in[pz + off[k]];
  }
  // The plane a step adds to the window is fetched one step ahead (in the stack dtype: half the registers of the
  // window's float64), so its HBM latency runs under the previous plane's three passes instead of in front of this one's.
  T nxt[SL];
  {
    const size_t pz = (size_t)mz[z_begin + W - 1] * plane;
#pragma unroll
    for (int k = 0; k < SL; ++k) nxt[k] = in[pz + off[k]];
  }
  for (int zz = z_begin; zz < z_end; zz += U) {
    auto step = [&](auto pc) -> bool {
      constexpr int ph = decltype(pc)::value;          // ring phase: logical slot j -> physical (ph + j) % W
      const int z = zz + ph;
      if (z >= z_end) return false;                    // block-uniform
      {
#pragma unroll
        for (int k = 0; k < SL; ++k) win[k][(ph + W - 1) % W] = (WT)nxt[k];
        const size_t pz = (size_t)mz[z + W] * plane;   // (the border map reaches Z + 2R: valid for the last plane too)
#pragma unroll
        for (int k = 0; k < SL; ++k) nxt[k] = in[pz + off[k]];
      }
      // axis 0 on the halo tile
#pragma unroll
      for (int k = 0; k < SL; ++k) {
        const int pos = tid + NT * k;
        if (pos < NPOS) {
          double acc = (double)win[k][(ph + R) % W] * taps.w[0];
#pragma unroll
          for (int j = R; j >= 1; --j)
            acc = acc + ((double)win[k][(ph + R - j) % W] + (double)win[k][(ph + R + j) % W]) * taps.w[j];
          const int xx = pos / EY, yy = pos - xx * EY;
          A[xx][yy] = (LT)cvt<T>(acc);
        }
        if constexpr (!WIDE) {
#pragma unroll
          for (int j = 0; j < W - 1; ++j) win[k][j] = win[k][j + 1];
        }
      }
      __syncthreads();
      if constexpr (WIDE) {
        // Axes 1 and 2 with register windows as well: a thread converts each LDS value once and uses it for up to 2R+1
        // outputs (7 conversions and 7 LDS reads per output before — the kernel issued 69 VALU instructions per voxel
        // for 30 of arithmetic, and VALU issue is what it shares with the long pass on the other stream).
        // axis 1: runs of RUN1 outputs along x at one yy; lanes along yy (conflict-free rows)
        constexpr int RUN1 = 8, NR1 = TX / RUN1;
        static_assert(TX % RUN1 == 0 && NR1 * EY <= NT, "axis-1 runs must fit one round of the block");
        if (tid < NR1 * EY) {
          const int yy = tid % EY, xr = (tid / EY) * RUN1;
          double v[RUN1 + 2 * R];
#pragma unroll
          for (int j = 0; j < RUN1 + 2 * R; ++j) v[j] = (double)A[xr + j][yy];
#pragma unroll
          for (int o = 0; o < RUN1; ++o) {
            double acc = v[o + R] * taps.w[0];
#pragma unroll
            for (int j = R; j >= 1; --j) acc = acc + (v[o + R - j] + v[o + R + j]) * taps.w[j];
            B[xr + o][yy] = (LT)cvt<T>(acc);
          }
        }
        __syncthreads();
        // axis 2: runs of RUN2 outputs along y at one x, fetched as 16-byte pieces of the row (rows are 16-byte aligned)
        constexpr int RUN2 = 4, NR2 = TY / RUN2;
        static_assert(TX * NR2 == NT && (RUN2 + 2 * R + 3) / 4 * 4 <= EY + PADY - (TY - RUN2), "axis-2 runs: one per thread, reads inside the row");
        {
          const int x = tid / NR2, yr = (tid % NR2) * RUN2;
          float f[(RUN2 + 2 * R + 3) / 4 * 4];
#pragma unroll
          for (int q = 0; q < (RUN2 + 2 * R + 3) / 4; ++q) *reinterpret_cast<float4*>(&f[4 * q]) = *reinterpret_cast<const float4*>(&B[x][yr + 4 * q]);
          double v[RUN2 + 2 * R];
#pragma unroll
          for (int j = 0; j < RUN2 + 2 * R; ++j) v[j] = (double)f[j];
          T* po = out + (size_t)z * plane + (size_t)(x0 + x) * Y + (y0 + yr);
#pragma unroll
          for (int o = 0; o < RUN2; ++o) {
            double acc = v[o + R] * taps.w[0];
#pragma unroll
            for (int j = R; j >= 1; --j) acc = acc + (v[o + R - j] + v[o + R + j]) * taps.w[j];
            if (x0 + x < X && y0 + yr + o < Y) po[o] = cvt<T>(acc);
          }
        }
        return true;
      }
      // axis 1: outputs (x, yy) for x in [0,TX), yy in [0,EY)
#pragma unroll
      for (int k = 0; k < S1; ++k) {
        const int o = tid + NT * k;
        if (o < NX1) {
          const int x = o / EY, yy = o - x * EY;
          double acc = (double)A[x + R][yy] * taps.w[0];
#pragma unroll
          for (int j = R; j >= 1; --j) acc = acc + ((double)A[x + R - j][yy] + (double)A[x + R + j][yy]) * taps.w[j];
          B[x][yy] = (LT)cvt<T>(acc);
        }
      }
      __syncthreads();
      // axis 2: outputs (x, y), lanes along y
      T* po = out + (size_t)z * plane;
#pragma unroll
      for (int k = 0; k < S2; ++k) {
        const int o = tid + NT * k;
        const int x = o / TY, y = o - x * TY;
        double acc = (double)B[x][y + R] * taps.w[0];
#pragma unroll
        for (int j = R; j >= 1; --j) acc = acc + ((double)B[x][y + R - j] + (double)B[x][y + R + j]) * taps.w[j];
        if (x0 + x < X && y0 + y < Y) po[(size_t)(x0 + x) * Y + (y0 + y)] = cvt<T>(acc);
      }
      // the next plane's A is written only after every thread has passed the barrier above (A is dead after axis 1),
      // B only after the next plane's first barrier: no third barrier needed
      return true;
    };
    static_for_until<0, U>(step);
  }
}

// ---- axes 1 and 2 of a short filter, plane by plane (the axis-0 pass was done by gauss_axis0_folded<.., RF>) --------
// One 256-thread block walks an x segment of one plane in steps of TXB rows over a y tile of TY <= 248 columns.
// Axis 1: thread t owns column y0 - R + t, loads TXB + 2R values of it (coalesced along y), converts each once and
// writes TXB outputs to LDS; axis 2: runs of four outputs along y from 16-byte LDS pieces, stored as one 16-byte piece.
// No z window, no halo planes: 60 registers, six blocks per CU.
template <class T, int R>
__global__ __launch_bounds__(256, 5) void gauss_xy_short(const T* __restrict__ in, T* __restrict__ out, int X, int Y, Taps taps,
                                                     const int* __restrict__ mx, const int* __restrict__ my, int TY, int xseg,
                                                     float* __restrict__ tmax) {
  constexpr int TXB = 16, RUN2 = 4, PITCH = 264, NV = (RUN2 + 2 * R + 3) / 4 * 4;
  __shared__ __attribute__((aligned(16))) float B[TXB][PITCH];
  __shared__ unsigned smx[4];   // per 64-column group of the tile: largest output of the step, as an order-preserving key
  const int t = threadIdx.x;
  // (y tile, x segment) pairs of a plane are dealt to the XCDs in runs, so that tiles sharing halo lines share an L2
  const int nty = (Y + TY - 1) / TY;
  const int tile = xcd_tile(blockIdx.x, nty * ((X + xseg - 1) / xseg));
  if (tile < 0) return;   // whole block
  const int y0 = (tile % nty) * TY, EY = TY + 2 * R;
  const int xbeg = (tile / nty) * xseg, xend = xbeg + xseg < X ? xbeg + xseg : X;
  // addressing: buffer descriptors with the row's byte offset as the scalar operand and a per-thread byte offset that is
  // fixed for the whole walk (the host checks that the stack is below 2^31 bytes): no vector address arithmetic per access
  const unsigned ES = (unsigned)sizeof(T);
  const unsigned pzb = (unsigned)blockIdx.z * (unsigned)X * (unsigned)Y * ES, rowb = (unsigned)Y * ES;
  const int nbytes = (int)((unsigned)gridDim.z * (unsigned)X * (unsigned)Y * ES);
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)in, (short)0, nbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)out, (short)0, nbytes, 0x00020000);
  const unsigned gyb = (t < EY ? (unsigned)my[y0 + t] : 0u) * ES;
  const int nr2 = TY / RUN2;                       // runs per row
  int rx[4], ry[4];
  unsigned so[4];                                  // byte offset of a run's first output inside the step's first row
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int id = t + 256 * r;
    rx[r] = id < TXB * nr2 ? id / nr2 : -1;
    ry[r] = (id % nr2) * RUN2;
    so[r] = ((unsigned)(rx[r] < 0 ? 0 : rx[r]) * (unsigned)Y + (unsigned)(y0 + ry[r])) * ES;
  }
  if (t < 4) smx[t] = 0u;   // (the first barrier of the loop orders this before the first atomic)
  const bool vec = (Y % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  // the column window slides by TXB rows per step; the TXB rows a step adds are fetched one step ahead (in the stack
  // dtype), so their latency runs under the previous step's two passes
  double v[TXB + 2 * R];
  T nxt[TXB];
  if (t < EY) {
#pragma unroll
    for (int j = 0; j < 2 * R; ++j) v[TXB + j] = (double)buf_ld<T>(rin, gyb, pzb + (unsigned)mx[xbeg + j] * rowb);
#pragma unroll
    for (int j = 0; j < TXB; ++j) nxt[j] = buf_ld<T>(rin, gyb, pzb + (unsigned)mx[xbeg + 2 * R + j] * rowb);
  }
  for (int xs = xbeg; xs < xend; xs += TXB) {
    if (t < EY) {
#pragma unroll
      for (int j = 0; j < 2 * R; ++j) v[j] = v[TXB + j];
#pragma unroll
      for (int j = 0; j < TXB; ++j) v[2 * R + j] = (double)nxt[j];
      if (xs + TXB < xend) {
#pragma unroll
        for (int j = 0; j < TXB; ++j) nxt[j] = buf_ld<T>(rin, gyb, pzb + (unsigned)mx[xs + TXB + 2 * R + j] * rowb);
      }
#pragma unroll
      for (int o = 0; o < TXB; ++o) {
        double acc = v[o + R] * taps.w[0];
#pragma unroll
        for (int j = R; j >= 1; --j) acc = acc + (v[o + R - j] + v[o + R + j]) * taps.w[j];
        B[o][t] = (float)cvt<T>(acc);
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int x = rx[r];
      float smax = -INFINITY;   // largest of this run's outputs
      if (x >= 0 && xs + x < xend) {
        float f[NV];
#pragma unroll
        for (int q = 0; q < NV / 4; ++q) *reinterpret_cast<float4*>(&f[4 * q]) = *reinterpret_cast<const float4*>(&B[x][ry[r] + 4 * q]);
        alignas(16) T res[RUN2];
#pragma unroll
        for (int o = 0; o < RUN2; ++o) {
          double acc = (double)f[o + R] * taps.w[0];
#pragma unroll
          for (int j = R; j >= 1; --j) acc = acc + ((double)f[o + R - j] + (double)f[o + R + j]) * taps.w[j];
          res[o] = cvt<T>(acc);
          smax = fmaxf(smax, (float)res[o]);
        }
        const int y = y0 + ry[r];
        const unsigned sb = pzb + (unsigned)xs * rowb;   // the step's first row (scalar)
        if (vec && y + RUN2 <= Y) {
          if constexpr (sizeof(T) == 4) __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const bv4u*>(res), rout, so[r], sb, 0);
          else __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const bv2u*>(res), rout, so[r], sb, 0);
        } else {
#pragma unroll
          for (int o = 0; o < RUN2; ++o) if (y + o < Y) buf_st<T>(res[o], rout, so[r] + (unsigned)o * ES, sb);
        }
      }
      if (tmax) {
        // TY is a multiple of 64 here: 16 consecutive runs (an aligned group of 16 lanes) are one row of one 64-column
        // group, i.e. of one detector tile; the group's maximum goes to its LDS slot
        // maximum over the 16 lanes of a DPP row: four rotations inside the row, fused into the max instructions
        smax = fmaxf(smax, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(smax), 0x128, 0xf, 0xf, false)));   // row_ror:8
        smax = fmaxf(smax, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(smax), 0x124, 0xf, 0xf, false)));   // row_ror:4
        smax = fmaxf(smax, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(smax), 0x122, 0xf, 0xf, false)));   // row_ror:2
        smax = fmaxf(smax, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(smax), 0x121, 0xf, 0xf, false)));   // row_ror:1
        if ((t & 15) == 0 && smax > -INFINITY) {
          const unsigned u = __float_as_uint(smax);
          atomicMax(&smx[ry[r] >> 6], (u & 0x80000000u) ? ~u : (u | 0x80000000u));
        }
      }
    }
    __syncthreads();
    if (tmax && t < (TY >> 6)) {   // [plane][step of TXB rows][64-column group]; the next step's atomics come behind its first barrier
      const int col = (y0 >> 6) + t, ncol = (Y + 63) >> 6;
      const unsigned k = smx[t];
      smx[t] = 0u;
      if (col < ncol)
        tmax[((size_t)blockIdx.z * ((X + TXB - 1) / TXB) + xs / TXB) * ncol + col] =
            k == 0u ? -INFINITY : __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
    }
  }
}

// KZ / KS / KC: outputs per chunk of the axis-0 pass, of the other strided passes, of the LDS-transposed pass
int g_cert = -2;   // -2: default guard (4R+8 ulps), -1: fused path off, >= 0: guard distance in ulps (tests)
inline int cert_for(int R) { return g_cert == -2 ? 4 * R + 8 : g_cert; }

int g_fold_on = 1;

// border map (position i - R -> source index) kept on the device per (count, R, len, mode): the plane-wise kernel is
// launched right behind the column kernel, a map-building launch in between would sit on the critical path
const int* cached_border_map(int count, int R, int len, int mode) {
  struct Key { int count, R, len, mode; const int* d; };
  static std::mutex mu;
  static std::vector<Key> maps;
  static pid_t owner_pid = 0;
  static int owner_dev = -1;
  std::lock_guard<std::mutex> g(mu);
  {   // device pointers of another process (fork) or another device mean nothing here: start over, as get_plan does
    int dev = -1;
    (void)hipGetDevice(&dev);
    if (owner_pid != getpid() || owner_dev != dev) { maps.clear(); owner_pid = getpid(); owner_dev = dev; }
  }
  for (const Key& k : maps)
    if (k.count == count && k.R == R && k.len == len && k.mode == mode) return k.d;
  std::vector<int> h(count);
  for (int i = 0; i < count; ++i) h[i] = border_idx(i - R, len, mode);
  int* d = nullptr;
  if (hipMalloc((void**)&d, (size_t)count * sizeof(int)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, h.data(), (size_t)count * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
  if (maps.size() >= 64) maps.erase(maps.begin());   // (the evicted map stays allocated: launches may be in flight)
  maps.push_back(Key{count, R, len, mode, d});
  return d;
}

template <class T>
int dog_pair_t(const T* src, int Z, int X, int Y, const Taps& ft, const Taps& bt, T* dst_front, T* dst_zp, T* tmp, hipStream_t s,
               float* tmax, float* smin, float* sabs) {
  constexpr int RF = 3, RB = 30;
  const size_t plane = (size_t)X * Y;
  bool nonneg = true;
  for (int j = 0; j <= RB; ++j) nonneg &= bt.w[j] >= 0.0;
  const int cert = !nonneg ? -1 : (g_cert == -2 ? 3 * RB + Z + 16 : g_cert);
  int rc;
  {
    ia3rt::ProfScope ps("gauss_axis0_pair");
    static_assert(RF == 3 && RB == 30, "the radii gauss_col.inc instantiates");
    if constexpr (std::is_same_v<T, float>) rc = folded_pair_f32(src, Z, plane, bt, dst_zp, ft, tmp, s, cert, smin, sabs, Y);
    else rc = folded_pair_u16(src, Z, plane, bt, dst_zp, ft, tmp, s, cert, smin, sabs, Y);
  }
  if (rc) return rc;   // FOLD_NOT_COVERED, or an error (negative)
  // axes 1 and 2 of the short filter: tmp -> dst_front (the caller goes on with dst_zp on the auxiliary stream)
  int TY, ntile;
  ia3k::dog_pair_tiles(X, Y, &TY, &ntile, nullptr);
  const int xseg = 128;   // a multiple of the 16-row step: steps start on multiples of 16
  // the candidate detector (seed_cand3_tiled) indexes the table of step maxima as [plane][row / 16][column / 64]: that is
  // what this launch writes only while a y tile holds whole 64-column groups and segments start on multiples of 16 rows
  if (TY % 64 != 0 || xseg % 16 != 0) return ia3rt::set_error(IA3_EUNSUPPORTED, "plane-wise filter geometry does not match the detector's tiles");
  const int cx = X + 2 * RF + 32, cy = ntile * TY + 2 * RF;
  const int* mx = cached_border_map(cx, RF, X, IA3_MODE_REFLECT);
  const int* my = cached_border_map(cy, RF, Y, IA3_MODE_REFLECT);
  if (!mx || !my) return ia3rt::set_error(IA3_ENOMEM, "border maps");
  // Fork point: what the caller queues on the auxiliary stream (buffer clears, the bound of the lazy background filter)
  // may start behind the column kernel.  The plane-wise filter itself — the long one of the two branches — stays on
  // the MAIN stream, back to back with the column kernel: a launch that waits for another stream's event starts ~20 us
  // later than one that follows its predecessor in the same queue, and the detector behind the join no longer waits
  // for the branch that ends last (rocprofv3 kernel trace of a bench step: 24 + 23 us of idle device -> 5 + 10).
  bool forked;
  { ia3rt::AuxScope aux; forked = aux.ok; }
  {
    ia3rt::ProfScope ps("gauss_xy_R3");
    const unsigned tiles = (unsigned)ntile * (unsigned)((X + xseg - 1) / xseg);
    dim3 g(8 * ((tiles + 7) / 8), 1, (unsigned)Z);   // tiles, XCD-grouped inside the kernel
    hipLaunchKernelGGL((gauss_xy_short<T, RF>), g, dim3(256), 0, s, (const T*)tmp, dst_front, X, Y, ft, mx, my, TY, xseg, tmax);
  }
  return forked ? 0 : FOLD_NO_FORK;
}

// axes: bit 0 = the axis-0 pass (src -> dst), bit 1 = the axis-1 and axis-2 passes (dst -> tmp -> dst)
template <class T, int R, int KS, int KC, int KZ = KS>
int run_fixed(const T* src, int Z, int X, int Y, const Taps& t, int mode, T* dst, T* tmp, hipStream_t s, int axes) {
  const size_t plane = (size_t)X * Y;
  if constexpr (R <= 3) {
    if (axes != 3) return ia3rt::set_error(IA3_EUNSUPPORTED, "the fused short filter runs all three axes at once");   // R = 6 was measured too: fused 2.26 ms, three passes 1.17 ms (2048x2048x50 f32)
    // maps must cover the halo of the last (partial) tile: positions up to ceil(len/tile)*tile + 2R
    static const std::string nf = "gauss_fused3_R" + std::to_string(R);
    ia3rt::ProfScope ps(nf.c_str());
    const int fz = Z + 2 * R + 1, fx = ((X + 31) / 32) * 32 + 2 * R, fy = ((Y + 63) / 64) * 64 + 2 * R;
    ia3rt::Scratch fm((size_t)(fz + fx + fy) * sizeof(int));
    if (!fm.p) return IA3_ENOMEM;
    int* qz = fm.as<int>();
    int* qx = qz + fz;
    int* qy = qx + fx;
    hipLaunchKernelGGL(border_map_k, dim3((fz + 255) / 256), dim3(256), 0, s, qz, fz, R, Z, mode);
    hipLaunchKernelGGL(border_map_k, dim3((fx + 255) / 256), dim3(256), 0, s, qx, fx, R, X, mode);
    hipLaunchKernelGGL(border_map_k, dim3((fy + 255) / 256), dim3(256), 0, s, qy, fy, R, Y, mode);
    const unsigned bx = (unsigned)((Y + 63) / 64), by = (unsigned)((X + 31) / 32);
    // split z when the (x, y) tiling alone cannot fill the chip (each z chunk re-reads 2R halo planes)
    int zseg = Z;
    while ((long long)bx * by * ((Z + zseg - 1) / zseg) < 1024 && zseg / 2 >= 4 * R) zseg = (zseg + 1) / 2;
    dim3 g(8 * ((bx * by + 7) / 8), 1, (unsigned)((Z + zseg - 1) / zseg));   // tiles, XCD-grouped inside the kernel
    hipLaunchKernelGGL((gauss3_fused<T, R>), g, dim3(fused_threads<R>()), 0, s, src, dst, Z, X, Y, t, (const int*)qz,
                       (const int*)qx, (const int*)qy, zseg);
    (void)tmp;
    return 0;
  }
  // the certified paths assume non-negative taps (caller-supplied weights may not be): reference sequence otherwise
  bool taps_nonneg = true;
  for (int j = 0; j <= R; ++j) taps_nonneg &= t.w[j] >= 0.0;
  const int cert = taps_nonneg ? cert_for(R) : -1;
  // border maps for the three axes: positions -R .. len + R + 2K (sliding-window prefetch overshoots by < 2K)
  const int cz = Z + 2 * R + 3 * KZ, cx = X + 2 * R + 3 * KS, cy = Y + 2 * R + 256 * 9;
  ia3rt::Scratch maps((size_t)(cz + cx + cy) * sizeof(int));
  if (!maps.p) return IA3_ENOMEM;
  int* mz = maps.as<int>();
  int* mx = mz + cz;
  int* my = mx + cx;
  hipLaunchKernelGGL(border_map_k, dim3((cz + 255) / 256), dim3(256), 0, s, mz, cz, R, Z, mode);
  hipLaunchKernelGGL(border_map_k, dim3((cx + 255) / 256), dim3(256), 0, s, mx, cx, R, X, mode);
  hipLaunchKernelGGL(border_map_k, dim3((cy + 255) / 256), dim3(256), 0, s, my, cy, R, Y, mode);
  static const std::string nz = "gauss_axis0_R" + std::to_string(R), nx = "gauss_axis1_R" + std::to_string(R),
                           ny = "gauss_axis2_R" + std::to_string(R);
  // segment length along the filter axis: a multiple of K, long enough to amortise the 2R-deep window fill
  auto seg_for_k = [](int len, long long lines, int kk) {
    int seg = ((len + kk - 1) / kk) * kk;                 // whole line
    const int min_seg = ((8 * R + kk - 1) / kk) * kk;     // halo re-read <= 25 %
    while (lines * ((len + seg - 1) / seg) < 256LL * 256 * 8 && seg / 2 >= min_seg) seg = ((seg / 2 + kk - 1) / kk) * kk;
    return seg;
  };
  auto seg_for = [&](int len, long long lines) { return seg_for_k(len, lines, KS); };
  // axis 0: src -> dst
  if (axes & 1) {
    ia3rt::ProfScope ps(nz.c_str());
    bool done = false;
    if constexpr (R >= 16) {
      if (g_fold_on && cert >= 0 && (size_t)Z * plane * sizeof(T) < 0x7fffffffULL) {   // short stacks: the column-in-registers form (guard: 3R + Z + 3 ulps, see the kernel; 32-bit buffer offsets)
        const int fc = g_cert == -2 ? 3 * R + Z + 16 : cert;
        int rc = FOLD_NOT_COVERED;
        if constexpr (R == 30) {   // the depths instantiated (IA3_FOLD_DEPTHS, radius 30); other stacks take the sliding window
          if constexpr (std::is_same_v<T, float>) rc = folded_axis0_f32(src, Z, plane, t, mode, dst, s, fc);
          else rc = folded_axis0_u16(src, Z, plane, t, mode, dst, s, fc);
        }
        if (rc < 0) return rc;
        done = rc == 0;
      }
    }
    if (!done) {
      const int seg = seg_for_k(Z, (long long)plane, KZ);
      dim3 g((unsigned)((plane + 255) / 256), 1, (unsigned)((Z + seg - 1) / seg));
      hipLaunchKernelGGL((gauss_strided<T, R, KZ>), g, dim3(256), 0, s, src, dst, (int)plane, plane, Z, (size_t)0, t, (const int*)mz, seg, cert);
    }
  }
  if (!(axes & 2)) return 0;
  if constexpr (R >= 16) {
    // long filters are f64-VALU-bound: give the contiguous axis the register-window kernel too, by transposing
    // each plane on the way out of the axis-1 pass and again on the way out of the axis-2 pass
    {  // axis 1: dst -> tmp, written transposed per plane: tmp[z][y][x]
      ia3rt::ProfScope ps(nx.c_str());
      const int seg = seg_for(X, (long long)Y * Z);
      dim3 g((unsigned)((Y + 255) / 256), (unsigned)Z, (unsigned)((X + seg - 1) / seg));
      hipLaunchKernelGGL((gauss_strided<T, R, KS, true>), g, dim3(256), 0, s, (const T*)dst, tmp, Y, (size_t)Y, X, plane, t, (const int*)mx, seg, cert);
    }
    {  // axis 2: tmp[z][y][x] -> dst[z][x][y]: lanes along x, filter along y with stride X, transposed store
      ia3rt::ProfScope ps(ny.c_str());
      const int seg = seg_for(Y, (long long)X * Z);
      dim3 g((unsigned)((X + 255) / 256), (unsigned)Z, (unsigned)((Y + seg - 1) / seg));
      hipLaunchKernelGGL((gauss_strided<T, R, KS, true>), g, dim3(256), 0, s, (const T*)tmp, dst, X, (size_t)X, Y, plane, t, (const int*)my, seg, cert);
    }
    return 0;
  }
  // short filters are HBM-bound: keep every store row-coalesced
  // axis 1: dst -> tmp
  {
    ia3rt::ProfScope ps(nx.c_str());
    const int seg = seg_for(X, (long long)Y * Z);
    dim3 g((unsigned)((Y + 255) / 256), (unsigned)Z, (unsigned)((X + seg - 1) / seg));
    hipLaunchKernelGGL((gauss_strided<T, R, KS>), g, dim3(256), 0, s, (const T*)dst, tmp, Y, (size_t)Y, X, plane, t, (const int*)mx, seg, cert);
  }
  // axis 2: tmp -> dst
  {
    ia3rt::ProfScope ps(ny.c_str());
    const size_t rows = (size_t)Z * X;
    constexpr int KY = KC;
    int seg = ((Y + 31) / 32) * 32;
    const int min_seg = ((8 * R + 31) / 32) * 32;
    while ((long long)((rows + 63) / 64) * ((Y + seg - 1) / seg) < 256LL * 24 && seg / 2 >= min_seg) seg = ((seg / 2 + 31) / 32) * 32;
    dim3 g((unsigned)((rows + 63) / 64), (unsigned)((Y + seg - 1) / seg), 1);
    hipLaunchKernelGGL((gauss_contig<T, R, KY>), g, dim3(64), 0, s, (const T*)tmp, dst, Y, rows, t, (const int*)my, seg);
  }
  return 0;
}

template <class T>
int run_generic(const T* src, int Z, int X, int Y, const double* w_host, int R, int mode, T* dst, T* tmp,
                hipStream_t s, int axes) {
  const size_t plane = (size_t)X * Y;
  ia3rt::Scratch wd((size_t)(R + 1) * sizeof(double));
  if (!wd.p) return IA3_ENOMEM;
  if (hipMemcpyAsync(wd.p, w_host, (size_t)(R + 1) * sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess)
    return ia3rt::set_error(IA3_EHIP, "tap upload failed");
  const double* w = wd.as<double>();
  if (axes & 1) {
    dim3 g((unsigned)((plane + 255) / 256), 1, (unsigned)(Z < 64 ? Z : 64));
    hipLaunchKernelGGL((gauss_generic<T>), g, dim3(256), 0, s, src, dst, (int)plane, plane, Z, (size_t)0, (size_t)1, w, R, mode);
  }
  if (!(axes & 2)) return 0;
  {
    dim3 g((unsigned)((Y + 255) / 256), (unsigned)Z, (unsigned)(X < 64 ? X : 64));
    hipLaunchKernelGGL((gauss_generic<T>), g, dim3(256), 0, s, (const T*)dst, tmp, Y, (size_t)Y, X, plane, (size_t)1, w, R, mode);
  }
  {
    // lines = rows (z,x) contiguous in y: inner index = row, inner_stride = Y, filter stride 1
    size_t rows = (size_t)Z * X;
    dim3 g((unsigned)((rows + 255) / 256), 1, (unsigned)(Y < 64 ? Y : 64));
    hipLaunchKernelGGL((gauss_generic<T>), g, dim3(256), 0, s, (const T*)tmp, dst, (int)rows, (size_t)1, Y, (size_t)0, (size_t)Y, w, R, mode);
  }
  return 0;  // scratch reuse is stream-ordered (single library stream)
}

template <class T>
int gaussian3d_t(const T* src, int Z, int X, int Y, const double* w, int R, int mode, T* dst, T* tmp, int axes) {
  hipStream_t s = ia3rt::stream();
  // taps by offset: wj[j] = w[R + j] (symmetric)
  if (R <= 63) {
    Taps t;
    for (int j = 0; j < 64; ++j) t.w[j] = j <= R ? w[R + j] : 0.0;
    switch (R) {
      case 3:  return run_fixed<T, 3, 16, 16>(src, Z, X, Y, t, mode, dst, tmp, s, axes);
      case 6:  return run_fixed<T, 6, 16, 16>(src, Z, X, Y, t, mode, dst, tmp, s, axes);
      case 10: return run_fixed<T, 10, 12, 16>(src, Z, X, Y, t, mode, dst, tmp, s, axes);
      case 30: return run_fixed<T, 30, 8, 8, 6>(src, Z, X, Y, t, mode, dst, tmp, s, axes);
      default: break;
    }
  }
  std::vector<double> wj(R + 1);
  for (int j = 0; j <= R; ++j) wj[j] = w[R + j];
  return run_generic<T>(src, Z, X, Y, wj.data(), R, mode, dst, tmp, s, axes);
}

}  // namespace

namespace ia3k {

int gaussian3d(const void* src, int dtype, int Z, int X, int Y, const double* w, int radius, int mode,
               void* dst, void* tmp, int axes) {
  if (radius < 0 || !w) return ia3rt::set_error(IA3_EINVAL, "bad taps");
  for (int j = 1; j <= radius; ++j)
    if (w[radius + j] != w[radius - j]) return ia3rt::set_error(IA3_EUNSUPPORTED, "taps must be symmetric");
  if ((size_t)X * Y > 0x7fffffffULL) return ia3rt::set_error(IA3_EUNSUPPORTED, "plane too large");
  int rc;
  if (dtype == IA3_F32) rc = gaussian3d_t<float>((const float*)src, Z, X, Y, w, radius, mode, (float*)dst, (float*)tmp, axes);
  else rc = gaussian3d_t<uint16_t>((const uint16_t*)src, Z, X, Y, w, radius, mode, (uint16_t*)dst, (uint16_t*)tmp, axes);
  if (rc) return rc;
  IA3_KCHECK();
  return IA3_OK;
}

// The DoG pair of get_seeds on one stack: the short filter (wf, radius rf) completely -> dst_front, and the axis-0 pass
// of the long filter (wb, radius rb) -> dst_zp, both 'reflect'.  The two axis-0 passes share one launch and every load;
// the short filter's other two axes are queued on the auxiliary stream.  Returns 0 with *forked = 1 when the caller
// has to aux_join() before reading dst_front, 1 when this shape / these radii are not covered (nothing was queued).
// geometry of the plane-wise kernel: y tiles of TY columns, steps of 16 rows; *count = entries per plane of the table of
// step maxima (one per 16 rows x 64 columns = per tile of the candidate detector)
void dog_pair_tiles(int X, int Y, int* ty, int* ntile, size_t* count) {
  const int ncol = (Y + 63) / 64;              // 64-column groups = the detector's tile columns
  *ty = 64 * (ncol < 3 ? ncol : 3);            // a y tile holds whole groups: 192 columns (198 of 256 threads on axis 1)
  *ntile = (Y + *ty - 1) / *ty;
  if (count) *count = (size_t)((X + 15) / 16) * (size_t)ncol;
}

// strip minima of the column kernel: [DOG_PAIR_ZGROUPS][X][Y / 32] floats each (smallest value, largest magnitude of the
// long filter's axis-0 result over a group of planes, one row and 32 columns); 0 when the kernel cannot produce them
size_t dog_pair_strips(int X, int Y) { return Y % 32 == 0 ? (size_t)DOG_PAIR_ZGROUPS * X * (Y / 32) : 0; }

int gauss_dog_pair(const void* src, int dtype, int Z, int X, int Y, const double* wf, int rf, const double* wb, int rb,
                   void* dst_front, void* dst_zp, void* tmp, int* forked, float* tmax, float* smin, float* sabs) {
  if (Y % 32 != 0) smin = sabs = nullptr;   // strips are aligned groups of 32 lanes (callers check dog_pair_strips first)
  *forked = 0;
  if (!g_fold_on || rf != 3 || rb != 30 || !fold_depth(Z) || (size_t)Z * X * Y * (dtype == IA3_F32 ? 4 : 2) >= 0x7fffffffULL || Y < 8 || X < 4) return 1;
  for (int j = 1; j <= rf; ++j) if (wf[rf + j] != wf[rf - j]) return 1;
  for (int j = 1; j <= rb; ++j) if (wb[rb + j] != wb[rb - j]) return 1;
  Taps ft, bt;
  for (int j = 0; j < 64; ++j) { ft.w[j] = j <= rf ? wf[rf + j] : 0.0; bt.w[j] = j <= rb ? wb[rb + j] : 0.0; }
  hipStream_t s = ia3rt::stream();
  int rc;
  if (dtype == IA3_F32) rc = dog_pair_t<float>((const float*)src, Z, X, Y, ft, bt, (float*)dst_front, (float*)dst_zp, (float*)tmp, s, tmax, smin, sabs);
  else rc = dog_pair_t<uint16_t>((const uint16_t*)src, Z, X, Y, ft, bt, (uint16_t*)dst_front, (uint16_t*)dst_zp, (uint16_t*)tmp, s, tmax, smin, sabs);
  if (rc < 0) return rc;                       // IA3 error codes are negative: the filtered stacks were not produced
  if (rc == FOLD_NOT_COVERED) return 1;
  *forked = rc == 0;                           // FOLD_NO_FORK: queued on the main stream, nothing to join
  IA3_KCHECK();
  return IA3_OK;
}

int highpass_combine(const void* im, const void* low, int dtype, size_t n, void* out) {
  hipStream_t s = ia3rt::stream();
  ia3rt::ProfScope ps("highpass_combine");
  unsigned blocks = (unsigned)((n + 255) / 256);
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (dtype == IA3_F32)
    hipLaunchKernelGGL((highpass_k<float>), dim3(blocks), dim3(256), 0, s, (const float*)im, (const float*)low, (float*)out, n);
  else
    hipLaunchKernelGGL((highpass_k<uint16_t>), dim3(blocks), dim3(256), 0, s, (const uint16_t*)im, (const uint16_t*)low, (uint16_t*)out, n);
  IA3_KCHECK();
  return IA3_OK;
}

}  // namespace ia3k

using namespace ia3rt;

extern "C" {

// see include/ia3.h
int ia3_prepare_depth(int dtype, int Z) {
  int rc = ensure_init(); if (rc) return rc;
  if (dtype != IA3_F32 && dtype != IA3_U16) return set_error(IA3_EINVAL, "dtype");
  return ia3g::column_kernel_source(dtype == IA3_F32, Z);
}
int ia3_set_tuning(int key, int value) {
  if (key == IA3_TUNE_GAUSS_CERT) {
    if (value < -2) return set_error(IA3_EINVAL, "IA3_TUNE_GAUSS_CERT: value must be >= -2");
    g_cert = value;
    return 0;
  }
  if (key == IA3_TUNE_GAUSS_FOLD) { g_fold_on = value != 0; return 0; }
  if (key == IA3_TUNE_DFT_VALU) { ia3k::set_dft_valu(value); return 0; }
  if (key == IA3_TUNE_UPLOAD_THREADS) return ia3rt::set_upload_threads(value);
  if (key == IA3_TUNE_SEED_DENSE) { ia3k::set_seed_dense(value); return 0; }
  if (key == IA3_TUNE_SEED_STRIPS) { ia3k::set_seed_strips(value); return 0; }
  if (key == IA3_TUNE_FIT_NBLIST) { ia3k::set_fit_nblist(value); return 0; }
  if (key == IA3_TUNE_FFT_C2C) { ia3k::set_fft_c2c(value); return 0; }
  if (key == IA3_TUNE_FIT_FUSE) { ia3k::set_fit_fuse(value); return 0; }
  if (key == IA3_TUNE_FIT_WAVES) { ia3k::set_fit_waves(value); return 0; }
  if (key == IA3_TUNE_FIT_MERGE) { ia3k::set_fit_merge(value); return 0; }
  if (key == IA3_TUNE_WARP_ONEPASS) { ia3k::set_warp_onepass(value); return 0; }
  if (key == IA3_TUNE_FIT_KDQ) { ia3k::set_fit_kdq(value); return 0; }
  if (key == IA3_TUNE_FIT_MEMO) { ia3k::set_fit_memo(value); return 0; }
  if (key == IA3_DEBUG_FIT_MAXFEV) { ia3k::set_fit_maxfev(value); return 0; }
  if (key == IA3_DEBUG_FIT_WAITBOUND) return ia3k::set_fit_waitbound(value);
  return set_error(IA3_EINVAL, "unknown tuning key");
}

static int taps_or_default(double sigma, double truncate, const double* weights, int radius,
                           std::vector<double>& w, int& R) {
  if (weights) {
    if (radius < 0) return set_error(IA3_EINVAL, "negative radius");
    w.assign(weights, weights + 2 * radius + 1);
    R = radius;
  } else {
    if (!(sigma > 0)) return set_error(IA3_EINVAL, "sigma must be > 0");
    gaussian_taps(sigma, truncate, w, R);
  }
  return IA3_OK;
}

int ia3_gaussian_filter_dev(const ia3_stack* im, double sigma, double truncate, int mode,
                            const double* weights, int radius, ia3_stack* out) {
  int rc = ensure_init(); if (rc) return rc;
  if (!im || !out) return set_error(IA3_EINVAL, "null stack");
  if (im->dtype != out->dtype || im->Z != out->Z || im->X != out->X || im->Y != out->Y)
    return set_error(IA3_EINVAL, "input/output stacks differ in shape or dtype");
  if (mode != IA3_MODE_REFLECT && mode != IA3_MODE_NEAREST) return set_error(IA3_EUNSUPPORTED, "mode %d", mode);
  std::vector<double> w; int R;
  rc = taps_or_default(sigma, truncate, weights, radius, w, R); if (rc) return rc;
  Scratch tmp(im->bytes);
  if (!tmp.p) return IA3_ENOMEM;
  if (im->d == out->d) return set_error(IA3_EINVAL, "in-place filtering is not supported");
  return ia3k::gaussian3d(im->d, im->dtype, im->Z, im->X, im->Y, w.data(), R, mode, out->d, tmp.p);
}

int ia3_gaussian_filter(const void* im, int dtype, int Z, int X, int Y, double sigma, double truncate,
                        int mode, const double* weights, int radius, void* out) {
  ia3_stack *a = nullptr, *b = nullptr;
  int rc = ia3_stack_upload(im, dtype, Z, X, Y, &a); if (rc) return rc;
  rc = ia3_stack_alloc(dtype, Z, X, Y, &b);
  if (!rc) rc = ia3_gaussian_filter_dev(a, sigma, truncate, mode, weights, radius, b);
  if (!rc) rc = ia3_stack_download(b, out);
  ia3_stack_free(a); ia3_stack_free(b);
  return rc;
}

int ia3_gaussian_highpass_dev(const ia3_stack* im, double sigma, double truncate,
                              const double* weights, int radius, ia3_stack* out) {
  int rc = ensure_init(); if (rc) return rc;
  if (!im || !out) return set_error(IA3_EINVAL, "null stack");
  if (im->dtype != out->dtype || im->Z != out->Z || im->X != out->X || im->Y != out->Y)
    return set_error(IA3_EINVAL, "input/output stacks differ in shape or dtype");
  if (im->d == out->d) return set_error(IA3_EINVAL, "in-place filtering is not supported");
  std::vector<double> w; int R;
  rc = taps_or_default(sigma, truncate, weights, radius, w, R); if (rc) return rc;
  Scratch tmp(im->bytes), low(im->bytes);
  if (!tmp.p || !low.p) return IA3_ENOMEM;
  rc = ia3k::gaussian3d(im->d, im->dtype, im->Z, im->X, im->Y, w.data(), R, IA3_MODE_NEAREST, low.p, tmp.p);
  if (rc) return rc;
  rc = ia3k::highpass_combine(im->d, low.p, im->dtype, (size_t)im->Z * im->X * im->Y, out->d);
  return rc;  // scratch reuse is stream-ordered (single library stream)
}

int ia3_gaussian_highpass(const void* im, int dtype, int Z, int X, int Y, double sigma, double truncate,
                          const double* weights, int radius, void* out) {
  ia3_stack *a = nullptr, *b = nullptr;
  int rc = ia3_stack_upload(im, dtype, Z, X, Y, &a); if (rc) return rc;
  rc = ia3_stack_alloc(dtype, Z, X, Y, &b);
  if (!rc) rc = ia3_gaussian_highpass_dev(a, sigma, truncate, weights, radius, b);
  if (!rc) rc = ia3_stack_download(b, out);
  ia3_stack_free(a); ia3_stack_free(b);
  return rc;
}

}  // extern "C"
