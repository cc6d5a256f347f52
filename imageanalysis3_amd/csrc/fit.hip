// Batched 3-D Gaussian LM fitting for gfx950: one wavefront per seed ball.
//
// Reference: External/Fitting_v4.py:559-683 iter_fit_seed_points (+ GaussianFit :165-396).
//   firstfit : every seed fitted on ball ∩ image ∩ {voxels whose nearest seed is this one}, data =
//              the ORIGINAL image -> all fits independent -> one wave per seed.
//   repeatfit: sweeps over unconverged seeds in seed order on the full ball; data = image minus the
//              current reconstructions of the other seeds (in-place Gauss–Seidel in the reference,
//              :658-675).  Only seeds whose balls can overlap interact, so the seeds are split into
//              connected components of the overlap graph; a wave walks one component in seed order,
//              sweep after sweep, and components run in parallel.  The residual image `im_add` (a
//              float64 copy of the whole stack in the reference, 1.7 GB per FOV) is never
//              materialised: data(v) = im(v) - Σ_{j≠i} rec_j(v) is rebuilt from the neighbours'
//              parameter vectors, which is the same quantity up to f64 rounding order.
//
// Wave layout: the ball has <= 512 voxels (radius 5: 512) = 8 slots per lane.  Per LM evaluation a
// lane computes residual + float32-rounded Jacobian row for its slots (float64, as NumPy does for
// the reference under numpy>=2) and accumulates its share of JᵀJ (55) and Jᵀr (10); a 6-step
// butterfly sums across the wave.  The 10x10 trust-region algebra (ia3_lm.h) runs redundantly on
// all lanes on a per-wave LDS work area, so control flow stays wave-uniform.
//
// Roofline: ~0.2 kflop per voxel per evaluation, 2 KB gathered per fit -> f64-VALU-bound, not HBM
// or MFMA (a 10x10 normal matrix is far below an MFMA tile).
#include "ia3_rt.h"
#include "ia3_lm.h"
#include "ia3_init.h"
#include <algorithm>
#include <numeric>
#include <math.h>
#include <string.h>
#include <unordered_map>

using namespace ia3;

namespace {

constexpr int SLOTS = 8;  // voxel slots per lane
constexpr int MAXBALL = 64 * SLOTS;

struct SeedState {
  double x[NP];   // unconstrained parameters of the last successful fit
  double delta;   // delta_center that fit used (needed to rebuild its reconstruction)
  int success;    // GaussianFit.success of the last attempt
  int has_rec;    // a reconstruction exists (ims_rec[ic] is an array, not NaN)
};

struct FitArgs {
  const void* im; int dtype; int Z, X, Y;
  const double* seeds;      // n x 3
  const int* nbr_off;       // n+1
  const int* nbr_idx;       // neighbours with |c_i - c_j|² <= (2r)², ascending
  const signed char* ball;  // nball x 4 (dz,dx,dy,0), np.indices order
  int nball, radius;
  SeedState* state;
  float* ps;                // n x 11
  int* nvox;                // n
  int* nfev;                // n (accumulated function evaluations)
  unsigned char* conv;      // n
  int* n_iter;              // max sweeps over components
  unsigned long long* counters;  // [0] fits run, [1] function evaluations
  double min_ws, max_ws, init_w, delta_first, delta_repeat, dist_th2;
  int n_max_iter;
  double ftol, xtol, gtol; int maxfev; double factor;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) { double o = __shfl_xor(v, m); v = o < v ? o : v; }
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) { double o = __shfl_xor(v, m); v = o > v ? o : v; }
  return v;
}

__device__ __forceinline__ double load_voxel(const void* im, int dtype, size_t idx) {
  return dtype == IA3_F32 ? (double)((const float*)im)[idx] : (double)((const uint16_t*)im)[idx];
}

// Per-lane view of one ball: up to SLOTS voxels.
struct Ball {
  float dat[SLOTS];                        // float32 data the fit sees (GaussianFit casts to float32, :172)
  float cz[SLOTS], cx[SLOTS], cy[SLOTS];   // voxel coordinates (exact small integers)
  unsigned valid;                          // bit s: slot s holds a voxel
};

// Wave-parallel evaluation of |f|, JᵀJ, Jᵀf for lm_solve.
struct WaveEval {
  const Ball* b;
  FitCfg cfg;
  __device__ double eval(const double* x, double* A, double* g) {
    Geom gm;
    make_geom(x, cfg, gm);
    double a[NTRI], gg[NP], ss = 0.0;
#pragma unroll
    for (int k = 0; k < NTRI; ++k) a[k] = 0.0;
#pragma unroll
    for (int k = 0; k < NP; ++k) gg[k] = 0.0;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      if (b->valid & (1u << s)) {
        double J[NP];
        double F = model_jac(gm, (double)b->cz[s], (double)b->cx[s], (double)b->cy[s], J);
        double r = (gm.ebk_f + F) - (double)b->dat[s];
        ss += r * r;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          gg[i] += J[i] * r;
#pragma unroll
          for (int j = i; j < NP; ++j) a[tri(i, j)] += J[i] * J[j];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NTRI; ++k) A[k] = wave_sum(a[k]);
#pragma unroll
    for (int k = 0; k < NP; ++k) g[k] = wave_sum(gg[k]);
    return sqrt(wave_sum(ss));
  }
};

// The ten smallest and ten largest of the wave's valid values, each ascending.
__device__ __forceinline__ void wave_extremes(const double* v, unsigned valid, double* lo10, double* hi10) {
  const int lane = threadIdx.x & 63;
  unsigned taken_lo = 0, taken_hi = 0;
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    double bl = INFINITY, bh = -INFINITY;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      bool ok = valid & (1u << s);
      if (ok && !(taken_lo & (1u << s)) && v[s] < bl) bl = v[s];
      if (ok && !(taken_hi & (1u << s)) && v[s] > bh) bh = v[s];
    }
    const double gl = wave_min(bl), gh = wave_max(bh);
    const unsigned long long ml = __ballot(bl == gl), mh = __ballot(bh == gh);
    if (lane == __ffsll((long long)ml) - 1) {
      bool done = false;
#pragma unroll
      for (int s = 0; s < SLOTS; ++s)
        if (!done && (valid & (1u << s)) && !(taken_lo & (1u << s)) && v[s] == gl) { taken_lo |= 1u << s; done = true; }
    }
    if (lane == __ffsll((long long)mh) - 1) {
      bool done = false;
#pragma unroll
      for (int s = 0; s < SLOTS; ++s)
        if (!done && (valid & (1u << s)) && !(taken_hi & (1u << s)) && v[s] == gh) { taken_hi |= 1u << s; done = true; }
    }
    lo10[k] = gl;
    hi10[9 - k] = gh;
  }
}

// One GaussianFit(...).fit() on the ball held by the wave (n >= 10 checked by the caller).
// vals: float64 data before the float32 cast (used for the start point only, :175-182).
__device__ __forceinline__ int wave_gaussfit(const FitArgs& fa, LMWork& w, const Ball& ball, const double* vals,
                                             int kind, const double* c0, double delta, int n, float* p_out) {
  WaveEval ev;
  ev.b = &ball;
  ev.cfg.min_ws = fa.min_ws; ev.cfg.max_ws = fa.max_ws; ev.cfg.delta = delta; ev.cfg.init_w = fa.init_w;
  ev.cfg.c0[0] = c0[0]; ev.cfg.c0[1] = c0[1]; ev.cfg.c0[2] = c0[2];
  double lo10[10], hi10[10];
  wave_extremes(vals, ball.valid, lo10, hi10);
  init_guess(lo10, hi10, kind, ev.cfg, w.x);
  LMResult r = lm_solve(ev, w, fa.ftol, fa.xtol, fa.gtol, fa.maxfev, fa.factor);
  to_natural(w.x, ev.cfg, p_out);
  Geom gm;
  make_geom(w.x, ev.cfg, gm);
  double s = 0.0;
#pragma unroll
  for (int sl = 0; sl < SLOTS; ++sl)
    if (ball.valid & (1u << sl))
      s += fabs((gm.ebk_f + model_f0(gm, (double)ball.cz[sl], (double)ball.cx[sl], (double)ball.cy[sl])) -
                (double)ball.dat[sl]);
  p_out[10] = (float)(wave_sum(s) / (double)n);
  return r.nfev;
}

__device__ __forceinline__ void store_result(const FitArgs& fa, int i, const float* p, const LMWork& w,
                                             double delta, bool ok, int n, int nfev) {
  if ((threadIdx.x & 63) == 0) {
    fa.nvox[i] = n;
    fa.nfev[i] += nfev;
    SeedState& st = fa.state[i];
    st.success = ok ? 1 : 0;
    if (ok) {
      for (int k = 0; k < NP; ++k) st.x[k] = w.x[k];
      st.delta = delta;
      st.has_rec = 1;
      for (int k = 0; k < 11; ++k) fa.ps[(size_t)i * 11 + k] = p[k];
      atomicAdd(&fa.counters[0], 1ull);
      atomicAdd(&fa.counters[1], (unsigned long long)nfev);
    }
  }
}

// ---- firstfit: one wave per seed (Fitting_v4.py:590-639) ----------------------------------------
__global__ __launch_bounds__(64) void fit_first_k(FitArgs fa, int n_seeds) {
  __shared__ LMWork w;
  const int i = blockIdx.x;
  if (i >= n_seeds) return;
  const int lane = threadIdx.x & 63;
  const double c0[3] = {fa.seeds[3 * i], fa.seeds[3 * i + 1], fa.seeds[3 * i + 2]};
  const int iz = (int)c0[0], ix = (int)c0[1], iy = (int)c0[2];  // Python int(): toward zero
  const int nb0 = fa.nbr_off[i], nb1 = fa.nbr_off[i + 1];
  Ball ball;
  double vals[SLOTS];
  ball.valid = 0;
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int vi = lane + 64 * s;
    ball.dat[s] = 0.f; ball.cz[s] = 0.f; ball.cx[s] = 0.f; ball.cy[s] = 0.f; vals[s] = 0.0;
    if (vi < fa.nball) {
      const int z = iz + fa.ball[4 * vi], x = ix + fa.ball[4 * vi + 1], y = iy + fa.ball[4 * vi + 2];
      bool ok = z >= 0 && z < fa.Z && x >= 0 && x < fa.X && y >= 0 && y < fa.Y;
      if (ok) {
        // Voronoi (:612, :422-424): drop the voxel if another seed is strictly nearer, or equally
        // near with a lower index (the reference's cKDTree leaves exact ties to its tree layout).
        const double dz = z - c0[0], dx = x - c0[1], dy = y - c0[2];
        const double dme = dz * dz + dx * dx + dy * dy;
        for (int q = nb0; q < nb1 && ok; ++q) {
          const int j = fa.nbr_idx[q];
          const double ez = z - fa.seeds[3 * j], ex = x - fa.seeds[3 * j + 1], ey = y - fa.seeds[3 * j + 2];
          const double dj = ez * ez + ex * ex + ey * ey;
          if (dj < dme || (dj == dme && j < i)) ok = false;
        }
      }
      if (ok) {
        const double v = load_voxel(fa.im, fa.dtype, ((size_t)z * fa.X + x) * fa.Y + y);
        ball.valid |= 1u << s;
        ball.dat[s] = (float)v; vals[s] = v;
        ball.cz[s] = (float)z; ball.cx[s] = (float)x; ball.cy[s] = (float)y;
      }
    }
  }
  const int n = (int)(wave_sum((double)__popc(ball.valid)) + 0.5);
  float p[11];
  int nfev = 0;
  const bool ok = n >= NP;  // :382-383
  if (ok) nfev = wave_gaussfit(fa, w, ball, vals, fa.dtype == IA3_F32 ? 0 : 1, c0, fa.delta_first, n, p);
  store_result(fa, i, p, w, fa.delta_first, ok, n, nfev);
}

// ---- repeatfit: one wave per connected component of the ball-overlap graph (:641-683) -----------
__global__ __launch_bounds__(64) void fit_repeat_k(FitArgs fa, const int* __restrict__ comp_off,
                                                   const int* __restrict__ comp_mem, int n_comp) {
  __shared__ LMWork w;
  const int c = blockIdx.x;
  if (c >= n_comp) return;
  const int lane = threadIdx.x & 63;
  const int m0 = comp_off[c], m1 = comp_off[c + 1];
  const int r = fa.radius;
  int sweeps = 0;
  bool all_conv;
  do {
    all_conv = true;
    for (int m = m0; m < m1; ++m) {
      const int i = comp_mem[m];
      if (fa.conv[i]) continue;
      const double c0[3] = {fa.seeds[3 * i], fa.seeds[3 * i + 1], fa.seeds[3 * i + 2]};
      const int iz = (int)c0[0], ix = (int)c0[1], iy = (int)c0[2];
      Ball ball;
      double vals[SLOTS];
      ball.valid = 0;
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        const int vi = lane + 64 * s;
        ball.dat[s] = 0.f; ball.cz[s] = 0.f; ball.cx[s] = 0.f; ball.cy[s] = 0.f; vals[s] = 0.0;
        if (vi < fa.nball) {
          const int z = iz + fa.ball[4 * vi], x = ix + fa.ball[4 * vi + 1], y = iy + fa.ball[4 * vi + 2];
          if (z >= 0 && z < fa.Z && x >= 0 && x < fa.X && y >= 0 && y < fa.Y) {
            ball.valid |= 1u << s;
            vals[s] = load_voxel(fa.im, fa.dtype, ((size_t)z * fa.X + x) * fa.Y + y);
            ball.cz[s] = (float)z; ball.cx[s] = (float)x; ball.cy[s] = (float)y;
          }
        }
      }
      // subtract the current reconstructions of the overlapping seeds (= im_add + own rec, :658-662)
      for (int q = fa.nbr_off[i]; q < fa.nbr_off[i + 1]; ++q) {
        const int j = fa.nbr_idx[q];
        const SeedState& sj = fa.state[j];
        if (!sj.has_rec) continue;
        FitCfg cj;
        cj.min_ws = fa.min_ws; cj.max_ws = fa.max_ws; cj.delta = sj.delta; cj.init_w = fa.init_w;
        cj.c0[0] = fa.seeds[3 * j]; cj.c0[1] = fa.seeds[3 * j + 1]; cj.c0[2] = fa.seeds[3 * j + 2];
        const int jz = (int)cj.c0[0], jx = (int)cj.c0[1], jy = (int)cj.c0[2];
        double xj[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) xj[k] = sj.x[k];
        Geom gj;
        make_geom(xj, cj, gj);
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
          if (ball.valid & (1u << s)) {
            const int oz = (int)ball.cz[s] - jz, ox = (int)ball.cx[s] - jx, oy = (int)ball.cy[s] - jy;
            if (oz >= -r && oz < r && ox >= -r && ox < r && oy >= -r && oy < r &&
                oz * oz + ox * ox + oy * oy <= r * r)
              vals[s] -= model_f0(gj, (double)ball.cz[s], (double)ball.cx[s], (double)ball.cy[s]);
          }
        }
      }
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) ball.dat[s] = (float)vals[s];
      const int n = (int)(wave_sum((double)__popc(ball.valid)) + 0.5);
      const int success_old = fa.state[i].success;
      const float co0 = fa.ps[(size_t)i * 11 + 1], co1 = fa.ps[(size_t)i * 11 + 2], co2 = fa.ps[(size_t)i * 11 + 3];
      float p[11];
      int nfev = 0;
      const bool ok = n >= NP;
      if (ok) nfev = wave_gaussfit(fa, w, ball, vals, 2, c0, fa.delta_repeat, n, p);
      store_result(fa, i, p, w, fa.delta_repeat, ok, n, nfev);
      // convergence (:677-680): float32 centre differences, compared in float64
      bool cv = true;
      if (ok && success_old) {
        const float d0 = co0 - p[1], d1 = co1 - p[2], d2 = co2 - p[3];
        const float dist = (d0 * d0 + d1 * d1) + d2 * d2;
        cv = (double)dist < fa.dist_th2;
      }
      if (lane == 0) fa.conv[i] = cv ? 1 : 0;
      all_conv = all_conv && cv;
      __threadfence();  // this wave re-reads state/ps/conv of its own component from memory
    }
    ++sweeps;
  } while (!all_conv && sweeps <= fa.n_max_iter);
  if (lane == 0) atomicMax(fa.n_iter, sweeps);
}

// ---- standalone GaussianFit(im, X, center).fit() on explicit voxel lists (Fitting_v4.py:165-396) --
struct VoxArgs {
  const double* vals;   // concatenated voxel values (float64 view of the caller's array)
  const int* coords;    // concatenated (z,x,y) triples
  const int* off;       // n_fits+1 offsets into vals / coords
  const double* center; // n_fits x 3
  const double* cfg;    // n_fits x 4: delta_center, min_w, max_w, init_w
  const int* kind;      // n_fits: 0 float32 data, 1 integer data, 2 float64 data (start-point arithmetic)
  float* ps;            // n_fits x 11
  double* xs;           // n_fits x 10 unconstrained solution
  int* info;            // n_fits x 2: success, nfev
};

__global__ __launch_bounds__(64) void fit_voxels_k(VoxArgs va, int n_fits, double ftol, double xtol, double gtol,
                                                   int maxfev, double factor) {
  __shared__ LMWork w;
  const int i = blockIdx.x;
  if (i >= n_fits) return;
  const int lane = threadIdx.x & 63;
  const int o0 = va.off[i], n = va.off[i + 1] - o0;
  Ball ball;
  double vals[SLOTS];
  ball.valid = 0;
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int vi = lane + 64 * s;
    ball.dat[s] = 0.f; ball.cz[s] = 0.f; ball.cx[s] = 0.f; ball.cy[s] = 0.f; vals[s] = 0.0;
    if (vi < n) {
      ball.valid |= 1u << s;
      vals[s] = va.vals[o0 + vi];
      ball.dat[s] = (float)vals[s];
      ball.cz[s] = (float)va.coords[3 * (o0 + vi)];
      ball.cx[s] = (float)va.coords[3 * (o0 + vi) + 1];
      ball.cy[s] = (float)va.coords[3 * (o0 + vi) + 2];
    }
  }
  FitArgs fa;
  const double min_w = va.cfg[4 * i + 1], max_w = va.cfg[4 * i + 2];
  fa.min_ws = min_w * min_w; fa.max_ws = max_w * max_w; fa.init_w = va.cfg[4 * i + 3];
  fa.ftol = ftol; fa.xtol = xtol; fa.gtol = gtol; fa.maxfev = maxfev; fa.factor = factor;
  const double c0[3] = {va.center[3 * i], va.center[3 * i + 1], va.center[3 * i + 2]};
  float p[11];
#pragma unroll
  for (int k = 0; k < 11; ++k) p[k] = NAN;
  int nfev = 0;
  const bool ok = n >= NP;
  if (ok) nfev = wave_gaussfit(fa, w, ball, vals, va.kind[i], c0, va.cfg[4 * i], n, p);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 11; ++k) va.ps[(size_t)i * 11 + k] = p[k];
    for (int k = 0; k < NP; ++k) va.xs[(size_t)i * NP + k] = ok ? w.x[k] : NAN;
    va.info[2 * i] = ok ? 1 : 0;
    va.info[2 * i + 1] = nfev;
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
using namespace ia3rt;

struct ia3_fitter {
  const ia3_stack* im;
  ia3_fit_params prm;
  int n;
  int n_comp;
  int nball;
  void *d_seeds, *d_nbr_off, *d_nbr_idx, *d_ball, *d_state, *d_ps, *d_nvox, *d_nfev, *d_conv, *d_niter,
      *d_counters, *d_comp_off, *d_comp_mem;
  bool first_done;
};

namespace {

int build_ball(int r, std::vector<signed char>& ball) {
  ball.clear();
  for (int z = -r; z < r; ++z)
    for (int x = -r; x < r; ++x)
      for (int y = -r; y < r; ++y)
        if (z * z + x * x + y * y <= r * r) { ball.push_back((signed char)z); ball.push_back((signed char)x); ball.push_back((signed char)y); ball.push_back(0); }
  return (int)(ball.size() / 4);
}

struct Cell { long long z, x, y; bool operator==(const Cell& o) const { return z == o.z && x == o.x && y == o.y; } };
struct CellHash { size_t operator()(const Cell& c) const { return (size_t)(c.z * 73856093LL ^ c.x * 19349663LL ^ c.y * 83492791LL); } };

// neighbours j != i with |c_i - c_j|² <= (2r)², ascending; connected components of that graph
void build_graph(const double* c, int n, double r2, std::vector<int>& off, std::vector<int>& idx,
                 std::vector<int>& comp_off, std::vector<int>& comp_mem) {
  const double cell = sqrt(r2) > 0 ? sqrt(r2) : 1.0;
  std::unordered_map<Cell, std::vector<int>, CellHash> grid;
  grid.reserve((size_t)n * 2);
  auto key = [&](int i) { return Cell{(long long)floor(c[3 * i] / cell), (long long)floor(c[3 * i + 1] / cell), (long long)floor(c[3 * i + 2] / cell)}; };
  for (int i = 0; i < n; ++i) grid[key(i)].push_back(i);
  off.assign(n + 1, 0);
  idx.clear();
  std::vector<int> parent(n);
  std::iota(parent.begin(), parent.end(), 0);
  auto find = [&](int a) { while (parent[a] != a) { parent[a] = parent[parent[a]]; a = parent[a]; } return a; };
  std::vector<int> tmp;
  for (int i = 0; i < n; ++i) {
    tmp.clear();
    Cell k = key(i);
    for (long long dz = -1; dz <= 1; ++dz)
      for (long long dx = -1; dx <= 1; ++dx)
        for (long long dy = -1; dy <= 1; ++dy) {
          auto it = grid.find(Cell{k.z + dz, k.x + dx, k.y + dy});
          if (it == grid.end()) continue;
          for (int j : it->second) {
            if (j == i) continue;
            double a = c[3 * i] - c[3 * j], b = c[3 * i + 1] - c[3 * j + 1], d = c[3 * i + 2] - c[3 * j + 2];
            if (a * a + b * b + d * d <= r2) tmp.push_back(j);
          }
        }
    std::sort(tmp.begin(), tmp.end());
    for (int j : tmp) { idx.push_back(j); int ra = find(i), rb = find(j); if (ra != rb) parent[ra > rb ? ra : rb] = ra > rb ? rb : ra; }
    off[i + 1] = (int)idx.size();
  }
  // components: members ascending; big components first (they are the serial tail of repeatfit)
  std::vector<std::vector<int>> comps;
  std::vector<int> cid(n, -1);
  for (int i = 0; i < n; ++i) {
    int rt = find(i);
    if (cid[rt] < 0) { cid[rt] = (int)comps.size(); comps.emplace_back(); }
    comps[cid[rt]].push_back(i);
  }
  std::stable_sort(comps.begin(), comps.end(), [](const std::vector<int>& a, const std::vector<int>& b) { return a.size() > b.size(); });
  comp_off.assign(1, 0);
  comp_mem.clear();
  for (auto& cm : comps) { comp_mem.insert(comp_mem.end(), cm.begin(), cm.end()); comp_off.push_back((int)comp_mem.size()); }
}

template <class T>
int dev_upload(void** d, const std::vector<T>& h) {
  size_t bytes = (h.size() ? h.size() : 1) * sizeof(T);
  if (hipMalloc(d, bytes) != hipSuccess) return set_error(IA3_ENOMEM, "hipMalloc(%zu) failed", bytes);
  if (h.size()) IA3_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return IA3_OK;
}
int dev_zero(void** d, size_t bytes) {
  if (bytes == 0) bytes = 8;
  if (hipMalloc(d, bytes) != hipSuccess) return set_error(IA3_ENOMEM, "hipMalloc(%zu) failed", bytes);
  IA3_HIP(hipMemset(*d, 0, bytes));
  return IA3_OK;
}

FitArgs make_args(const ia3_fitter* f) {
  FitArgs a;
  a.im = f->im->d; a.dtype = f->im->dtype; a.Z = f->im->Z; a.X = f->im->X; a.Y = f->im->Y;
  a.seeds = (const double*)f->d_seeds; a.nbr_off = (const int*)f->d_nbr_off; a.nbr_idx = (const int*)f->d_nbr_idx;
  a.ball = (const signed char*)f->d_ball; a.nball = f->nball; a.radius = f->prm.radius_fit;
  a.state = (SeedState*)f->d_state; a.ps = (float*)f->d_ps; a.nvox = (int*)f->d_nvox; a.nfev = (int*)f->d_nfev;
  a.conv = (unsigned char*)f->d_conv; a.n_iter = (int*)f->d_niter; a.counters = (unsigned long long*)f->d_counters;
  a.min_ws = f->prm.min_w * f->prm.min_w; a.max_ws = f->prm.max_w * f->prm.max_w; a.init_w = f->prm.init_w;
  a.delta_first = f->prm.min_delta_center; a.delta_repeat = f->prm.max_delta_center;
  a.dist_th2 = f->prm.max_dist_th * f->prm.max_dist_th;
  a.n_max_iter = f->prm.n_max_iter;
  // scipy.optimize.leastsq defaults used at Fitting_v4.py:388
  a.ftol = 1.49012e-8; a.xtol = 1.49012e-8; a.gtol = 0.0; a.maxfev = 1000; a.factor = 100.0;
  return a;
}

}  // namespace

extern "C" {

void ia3_fit_destroy(ia3_fitter* f) {
  if (!f) return;
  void* ptrs[] = {f->d_seeds, f->d_nbr_off, f->d_nbr_idx, f->d_ball, f->d_state, f->d_ps, f->d_nvox, f->d_nfev,
                  f->d_conv, f->d_niter, f->d_counters, f->d_comp_off, f->d_comp_mem};
  for (void* p : ptrs) if (p) hipFree(p);
  delete f;
}

int ia3_fit_create(const ia3_stack* im, const double* centers_zxy, int n, const ia3_fit_params* p,
                   ia3_fitter** out) {
  int rc = ensure_init(); if (rc) return rc;
  if (!im || !p || !out || n < 0 || (n > 0 && !centers_zxy)) return set_error(IA3_EINVAL, "bad argument");
  if (p->radius_fit < 1) return set_error(IA3_EINVAL, "radius_fit must be >= 1");
  std::vector<signed char> ball;
  int nball = build_ball(p->radius_fit, ball);
  if (nball > MAXBALL) return set_error(IA3_EUNSUPPORTED, "radius_fit %d gives %d voxels (> %d)", p->radius_fit, nball, MAXBALL);
  for (int i = 0; i < 3 * n; ++i)
    if (!(fabs(centers_zxy[i]) < 1e9)) return set_error(IA3_EINVAL, "non-finite seed coordinate");
  ia3_fitter* f = new ia3_fitter();
  memset(f, 0, sizeof(*f));
  f->im = im; f->prm = *p; f->n = n; f->nball = nball;
  std::vector<double> seeds(centers_zxy, centers_zxy + 3 * (size_t)n);
  std::vector<int> off, idx, coff, cmem;
  double rr = 2.0 * p->radius_fit;
  build_graph(seeds.data(), n, rr * rr, off, idx, coff, cmem);
  f->n_comp = (int)coff.size() - 1;
  rc = dev_upload(&f->d_seeds, seeds);
  if (!rc) rc = dev_upload(&f->d_nbr_off, off);
  if (!rc) rc = dev_upload(&f->d_nbr_idx, idx);
  if (!rc) rc = dev_upload(&f->d_ball, ball);
  if (!rc) rc = dev_upload(&f->d_comp_off, coff);
  if (!rc) rc = dev_upload(&f->d_comp_mem, cmem);
  if (!rc) rc = dev_zero(&f->d_state, sizeof(SeedState) * (size_t)n);
  if (!rc) rc = dev_zero(&f->d_nvox, sizeof(int) * (size_t)n);
  if (!rc) rc = dev_zero(&f->d_nfev, sizeof(int) * (size_t)n);
  if (!rc) rc = dev_zero(&f->d_conv, (size_t)n);
  if (!rc) rc = dev_zero(&f->d_niter, sizeof(int));
  if (!rc) rc = dev_zero(&f->d_counters, 2 * sizeof(unsigned long long));
  if (!rc) {
    std::vector<float> nanrows((size_t)n * 11, NAN);  // failed fits stay NaN rows (:636)
    rc = dev_upload(&f->d_ps, nanrows);
  }
  if (rc) { ia3_fit_destroy(f); return rc; }
  *out = f;
  return IA3_OK;
}

int ia3_fit_first(ia3_fitter* f) {
  if (!f) return set_error(IA3_EINVAL, "null fitter");
  if (f->n > 0) {
    FitArgs a = make_args(f);
    ProfScope ps("fit_first");
    hipLaunchKernelGGL(fit_first_k, dim3((unsigned)f->n), dim3(64), 0, stream(), a, f->n);
    IA3_KCHECK();
  }
  f->first_done = true;
  return IA3_OK;
}

int ia3_fit_repeat(ia3_fitter* f, int* n_iter) {
  if (!f) return set_error(IA3_EINVAL, "null fitter");
  if (!f->first_done) return set_error(IA3_EINVAL, "repeatfit() before firstfit()");
  int it = 0;
  if (f->n > 0) {
    FitArgs a = make_args(f);
    IA3_HIP(hipMemsetAsync(f->d_conv, 0, (size_t)f->n, stream()));
    IA3_HIP(hipMemsetAsync(f->d_niter, 0, sizeof(int), stream()));
    {
      ProfScope ps("fit_repeat");
      hipLaunchKernelGGL(fit_repeat_k, dim3((unsigned)f->n_comp), dim3(64), 0, stream(), a,
                         (const int*)f->d_comp_off, (const int*)f->d_comp_mem, f->n_comp);
    }
    IA3_KCHECK();
    IA3_HIP(hipMemcpyAsync(&it, f->d_niter, sizeof(int), hipMemcpyDeviceToHost, stream()));
    IA3_HIP(hipStreamSynchronize(stream()));
  }
  if (n_iter) *n_iter = it;
  return IA3_OK;
}

int ia3_fit_results(ia3_fitter* f, float* ps, uint8_t* success, int* nvox) {
  if (!f) return set_error(IA3_EINVAL, "null fitter");
  IA3_HIP(hipStreamSynchronize(stream()));
  if (f->n == 0) return IA3_OK;
  if (ps) IA3_HIP(hipMemcpy(ps, f->d_ps, sizeof(float) * 11 * (size_t)f->n, hipMemcpyDeviceToHost));
  if (nvox) IA3_HIP(hipMemcpy(nvox, f->d_nvox, sizeof(int) * (size_t)f->n, hipMemcpyDeviceToHost));
  if (success) {
    std::vector<SeedState> st(f->n);
    IA3_HIP(hipMemcpy(st.data(), f->d_state, sizeof(SeedState) * (size_t)f->n, hipMemcpyDeviceToHost));
    for (int i = 0; i < f->n; ++i) success[i] = (uint8_t)st[i].success;
  }
  return IA3_OK;
}

int ia3_fit_stats(ia3_fitter* f, int64_t* total_fits, int64_t* total_nfev) {
  if (!f) return set_error(IA3_EINVAL, "null fitter");
  unsigned long long c[2];
  IA3_HIP(hipStreamSynchronize(stream()));
  IA3_HIP(hipMemcpy(c, f->d_counters, sizeof(c), hipMemcpyDeviceToHost));
  if (total_fits) *total_fits = (int64_t)c[0];
  if (total_nfev) *total_nfev = (int64_t)c[1];
  return IA3_OK;
}

int ia3_fit_seeds(const void* im, int dtype, int Z, int X, int Y, const double* centers_zxy, int n,
                  const ia3_fit_params* p, float* out_ps, uint8_t* success, int* n_iter) {
  ia3_stack* s = nullptr;
  ia3_fitter* f = nullptr;
  int rc = ia3_stack_upload(im, dtype, Z, X, Y, &s); if (rc) return rc;
  rc = ia3_fit_create(s, centers_zxy, n, p, &f);
  if (!rc) rc = ia3_fit_first(f);
  if (!rc) rc = ia3_fit_repeat(f, n_iter);
  if (!rc) rc = ia3_fit_results(f, out_ps, success, nullptr);
  ia3_fit_destroy(f);
  ia3_stack_free(s);
  return rc;
}

// Batch of independent GaussianFit(im, X, center=...).fit() calls on explicit voxel lists (<= 512 voxels each).
// vals/coords are concatenated; off has n_fits+1 entries.  Outputs: ps n_fits x 11 (NaN rows when a fit
// has fewer than 10 voxels, Fitting_v4.py:382-383), xs n_fits x 10, info n_fits x 2 (success, nfev).
int ia3_gaussfit_voxels(const double* vals, const int* coords_zxy, const int* off, int n_fits, const double* centers,
                        const double* cfg4, const int* kind, float* ps, double* xs, int* info) {
  int rc = ensure_init(); if (rc) return rc;
  if (n_fits < 0 || (n_fits > 0 && (!vals || !coords_zxy || !off || !centers || !cfg4 || !kind || !ps)))
    return set_error(IA3_EINVAL, "bad argument");
  if (n_fits == 0) return IA3_OK;
  const int total = off[n_fits];
  for (int i = 0; i < n_fits; ++i)
    if (off[i + 1] - off[i] > MAXBALL || off[i + 1] < off[i])
      return set_error(IA3_EUNSUPPORTED, "fit %d has %d voxels (> %d)", i, off[i + 1] - off[i], MAXBALL);
  hipStream_t st = stream();
  Scratch dv((size_t)(total ? total : 1) * sizeof(double)), dc((size_t)(total ? total : 1) * 3 * sizeof(int)),
      doff((size_t)(n_fits + 1) * sizeof(int)), dcen((size_t)n_fits * 3 * sizeof(double)),
      dcfg((size_t)n_fits * 4 * sizeof(double)), dk((size_t)n_fits * sizeof(int)), dps((size_t)n_fits * 11 * sizeof(float)),
      dxs((size_t)n_fits * NP * sizeof(double)), dinfo((size_t)n_fits * 2 * sizeof(int));
  if (!dv.p || !dc.p || !doff.p || !dcen.p || !dcfg.p || !dk.p || !dps.p || !dxs.p || !dinfo.p) return IA3_ENOMEM;
  if (total) {
    IA3_HIP(hipMemcpyAsync(dv.p, vals, (size_t)total * sizeof(double), hipMemcpyHostToDevice, st));
    IA3_HIP(hipMemcpyAsync(dc.p, coords_zxy, (size_t)total * 3 * sizeof(int), hipMemcpyHostToDevice, st));
  }
  IA3_HIP(hipMemcpyAsync(doff.p, off, (size_t)(n_fits + 1) * sizeof(int), hipMemcpyHostToDevice, st));
  IA3_HIP(hipMemcpyAsync(dcen.p, centers, (size_t)n_fits * 3 * sizeof(double), hipMemcpyHostToDevice, st));
  IA3_HIP(hipMemcpyAsync(dcfg.p, cfg4, (size_t)n_fits * 4 * sizeof(double), hipMemcpyHostToDevice, st));
  IA3_HIP(hipMemcpyAsync(dk.p, kind, (size_t)n_fits * sizeof(int), hipMemcpyHostToDevice, st));
  VoxArgs va;
  va.vals = dv.as<double>(); va.coords = dc.as<int>(); va.off = doff.as<int>(); va.center = dcen.as<double>();
  va.cfg = dcfg.as<double>(); va.kind = dk.as<int>(); va.ps = dps.as<float>(); va.xs = dxs.as<double>(); va.info = dinfo.as<int>();
  {
    ProfScope ps_("fit_voxels");
    hipLaunchKernelGGL(fit_voxels_k, dim3((unsigned)n_fits), dim3(64), 0, st, va, n_fits, 1.49012e-8, 1.49012e-8, 0.0, 1000, 100.0);
  }
  IA3_KCHECK();
  IA3_HIP(hipMemcpyAsync(ps, dps.p, (size_t)n_fits * 11 * sizeof(float), hipMemcpyDeviceToHost, st));
  if (xs) IA3_HIP(hipMemcpyAsync(xs, dxs.p, (size_t)n_fits * NP * sizeof(double), hipMemcpyDeviceToHost, st));
  if (info) IA3_HIP(hipMemcpyAsync(info, dinfo.p, (size_t)n_fits * 2 * sizeof(int), hipMemcpyDeviceToHost, st));
  IA3_HIP(hipStreamSynchronize(st));
  return IA3_OK;
}

}  // extern "C"
