// Batched 3-D Gaussian LM fitting for gfx950: one wavefront per seed ball.
//
// Reference: External/Fitting_v4.py:559-683 iter_fit_seed_points (+ GaussianFit :165-396).
//   firstfit : every seed fitted on ball ∩ image ∩ {voxels whose nearest seed is this one}, data =
//              the ORIGINAL image -> all fits independent -> one wave per seed.
//   repeatfit: sweeps over unconverged seeds in seed order on the full ball; data = image minus the
//              current reconstructions of the other seeds (in-place Gauss–Seidel in the reference,
//              :658-675).  Only seeds whose balls can overlap interact: the work list (stages x seeds)
//              is walked by persistent waves that draw positions as tickets and wait for the fits a
//              position depends on (fit_stages_k), which reproduces the sequential order exactly; a
//              seed that overlaps no other gets its first fit and sweep 1 from one wave.  The residual
//              image `im_add` (a float64 copy of the whole stack in the reference, 1.7 GB per FOV) is
//              never materialised: data(v) = im(v) - Σ_{j≠i} rec_j(v) is rebuilt from the neighbours'
//              parameter vectors, which is the same quantity up to f64 rounding order.
//
// Wave layout: the ball has <= 512 voxels (radius 5: 512) = 8 slots per lane.  Per LM evaluation a
// lane computes residual + float32-rounded Jacobian row for its slots (float64, as NumPy does for
// the reference under numpy>=2) and accumulates its share of JᵀJ (55) and Jᵀr (10); permlane swaps
// and DPP rotations sum across the wave.  The 10x10 trust-region algebra (ia3_lm.h) runs redundantly on
// all lanes on a per-wave LDS work area, so control flow stays wave-uniform.
//
// Roofline: ~0.2 kflop per voxel per evaluation, 2 KB gathered per fit -> f64-VALU-bound, not HBM
// or MFMA (a 10x10 normal matrix is far below an MFMA tile).
#include "ia3_rt.h"
#include "ia3_lm.h"
#include "ia3_init.h"
#include "ia3_kdtree.h"
#include <algorithm>
#include <numeric>
#include <math.h>
#include <string.h>
#include <stddef.h>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <mutex>

using namespace ia3;

namespace ia3k {   // kdtree.cpp: the seed tree with scipy.spatial.cKDTree's layout
void kd_build(const double* points, int n, std::vector<ia3::KdNode>& nodes, std::vector<int>& indices, double* mins,
              double* maxes);
}

namespace {

constexpr int SLOTS = 8;  // voxel slots per lane
constexpr int MAXNB = 64; // neighbour slots per seed (seeds within 2r); denser seeds re-scan the seed list instead
constexpr int MAXBALL = 64 * SLOTS;

struct SeedState {
  double x[NP];   // unconstrained parameters of the last successful fit
  double delta;   // delta_center that fit used (needed to rebuild its reconstruction)
  int success;    // GaussianFit.success of the last attempt
  int has_rec;    // a reconstruction exists (ims_rec[ic] is an array, not NaN)
  int conv;       // converged flag of repeatfit (:680)
  int ver;        // how often this record has been rewritten (store_result with a fit made): what a refit of a NEIGHBOUR
                  // saw of this seed is unchanged as long as this number is
};

// One fitter may hold the seeds of SEVERAL fields of view (same shape and dtype; ia3_fit_fovs): the seeds of FOV f are
// the index range [fov_start[f], fov_start[f+1]), neighbours and dependencies exist only inside a range, and the work
// list runs over all of them — the long dependent chains and maxfev fits of one field overlap with the other fields' work.
struct FitArgs {
  const void* const* ims;   // n_fov stacks (device pointers)
  const int* fov_of;        // n: field of view of seed i
  const int* fov_start;     // n_fov + 1
  int n_fov;
  int dtype; int Z, X, Y;
  const double* seeds;      // n x 3
  int n;                    // number of seeds (all fields)
  int fuse;                 // a seed without neighbours: first fit and sweep 1 by the same wave (see fit_stages_k)
  int nb_cap;               // lists longer than this are not used (MAXNB; lowered by IA3_TUNE_FIT_NBLIST in tests)
  double nb_r2;             // (2r)²: seeds closer than this interact
  const int* nbr_cnt;       // n: number of neighbours of seed i (NOT clamped: > MAXNB = list overflow, see each_neighbour)
  const int* nbr_idx;       // n x MAXNB: the first MAXNB seeds j != i with |c_i - c_j|² <= (2r)², ascending
  const int* ball;          // nball packed offsets (dz, dx, dy, 0) as signed bytes, np.indices order: one load per voxel
  const unsigned long long* tie_lost;  // n x SLOTS lane masks: voxels this seed loses in an exact Voronoi tie (voronoi_ties_k); may be null
  int nball, radius;
  SeedState* state;
  float* ps;                // n x 11
  int* nvox;                // n
  int* nfev;                // n (accumulated function evaluations)
  unsigned char* conv;      // n
  // n: bit 63 = this seed has been refitted by a repeat sweep, low bits = sum of its neighbours' SeedState::ver at that
  // refit.  A repeat fit is a function of the image ball and of the neighbours' records alone (the start point is the
  // seed, Fitting_v4.py:664-666; the data the image minus the neighbours' reconstructions), so a sweep that finds the sum
  // unchanged would repeat the previous fit bit for bit: it is not run (stage_position).
  unsigned long long* memo;
  int* n_iter;              // n_fov: sweeps made per field (max over its seeds)
  unsigned long long* counters;  // [0] fits run, [1] function evaluations, [2] voxel evaluations (sum of nfev x voxels)
  unsigned long long* fov_counters;   // n_fov x 4: the same three per field; [3]: its seeds with neighbours (nbr_build_k)
  double min_ws, max_ws, init_w, delta_first, delta_repeat, dist_th2;
  int n_max_iter;
  double ftol, xtol, gtol; int maxfev; double factor;
  int variant = 0;            // FitCfg::variant (0 = Fitting_v4, 1 = legacy Fitting_v3)
  double iw[3] = {0, 0, 0};   // variant 1: start widths in w_ space
};

// ---- single-value wave reductions without LDS: four DPP rotations inside the 16-lane rows, then the two row /
// half-wave swaps of gfx950 (v_permlane16_swap / v_permlane32_swap with the value as both operands) -----------------
__device__ __forceinline__ double lane_mk(unsigned lo, unsigned hi) { return __hiloint2double((int)hi, (int)lo); }
template <int N>
__device__ __forceinline__ double lane_ror(double v) {   // value of the lane N places further along the same row
  // (mov_dpp, not update_dpp with an `old` operand: a row rotation reads a valid lane everywhere, and an explicit old
  // value costs two register initialisations per rotation — a third of the cross-lane sums' instructions)
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x120 | N, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x120 | N, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// the partner copies of v across the row pairs (0<->1, 2<->3) and across the half-waves
__device__ __forceinline__ void lane_swap16(double v, double& a, double& b) {
  auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
  auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
  a = lane_mk(lo[0], hi[0]); b = lane_mk(lo[1], hi[1]);   // a = even row of the pair, b = odd row, in every lane of the pair
}
__device__ __forceinline__ void lane_swap32(double v, double& a, double& b) {
  auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
  auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
  a = lane_mk(lo[0], hi[0]); b = lane_mk(lo[1], hi[1]);   // a = lower half's value, b = upper half's, in both halves
}
struct OpSum { __device__ __forceinline__ double operator()(double x, double y) const { return x + y; } };
struct OpMin { __device__ __forceinline__ double operator()(double x, double y) const { return y < x ? y : x; } };
struct OpMax { __device__ __forceinline__ double operator()(double x, double y) const { return y > x ? y : x; } };
template <class Op>
__device__ __forceinline__ double wave_reduce(double v, Op op) {
  v = op(v, lane_ror<8>(v));
  v = op(v, lane_ror<4>(v));
  v = op(v, lane_ror<2>(v));
  v = op(v, lane_ror<1>(v));
  double a, b;
  lane_swap16(v, a, b); v = op(a, b);
  lane_swap32(v, a, b); v = op(a, b);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) { return wave_reduce(v, OpSum()); }
__device__ __forceinline__ double wave_min(double v) { return wave_reduce(v, OpMin()); }
__device__ __forceinline__ double wave_max(double v) { return wave_reduce(v, OpMax()); }

// Cross-wave hand-off of per-seed results (persistent kernel below): every store of the handed-off words is a
// relaxed agent-scope atomic (write-through `sc1`), every load of them a relaxed agent-scope atomic (`sc1`, served
// by L2/memory, never by a stale L1 line); the producer drains its stores (vmcnt(0)) before it raises done[i].
template <class T> __device__ __forceinline__ T ld_sc1(const T* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#define LDH(ptr) ld_sc1(ptr)
template <class T> __device__ __forceinline__ void st_sc1(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ double load_voxel(const void* im, int dtype, size_t idx) {
  return dtype == IA3_F32 ? (double)((const float*)im)[idx] : (double)((const uint16_t*)im)[idx];
}

// A wave-uniform double as a scalar-register pair (the value must be the same in every lane).
__device__ __forceinline__ double sgpr(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// ---- cross-lane sums of many values at once (gfx950) -----------------------------------------------------------
// v_permlane32_swap / v_permlane16_swap exchange half-waves / odd-even rows of TWO registers in one VALU instruction,
// so one add folds two values by one butterfly level ("reduce-scatter"): 2N values -> N registers (xor 32) -> N/2
// registers (xor 16), each then holding four different values in its four 16-lane rows; the remaining four levels
// run inside the rows with DPP rotations.  65 values cost ~350 VALU instructions instead of 65 x 6 x (2 ds_bpermute +
// add) = 1170 with LDS round trips.
// lanes 0-31: x[l] + x[l+32];  lanes 32-63: y[l-32] + y[l]
__device__ __forceinline__ double swap32_add(double x, double y) {
  auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return lane_mk(lo[0], hi[0]) + lane_mk(lo[1], hi[1]);
}
// rows of 16 lanes: row0 = p.row0 + p.row1, row1 = q.row0 + q.row1, row2 = p.row2 + p.row3, row3 = q.row2 + q.row3
__device__ __forceinline__ double swap16_add(double p, double q) {
  auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(p), (unsigned)__double2loint(q), false, false);
  auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(p), (unsigned)__double2hiint(q), false, false);
  return lane_mk(lo[0], hi[0]) + lane_mk(lo[1], hi[1]);
}
__device__ __forceinline__ double row_sum16(double v) {   // every lane of a row ends with the row's total
  v = v + lane_ror<8>(v);
  v = v + lane_ror<4>(v);
  v = v + lane_ror<2>(v);
  v = v + lane_ror<1>(v);
  return v;
}
// The first two levels of row_sum16 for TWO registers at once, the way the half-wave swaps above fold the levels across
// rows: after v + ror8(v) lane l and lane l ^ 8 of a row hold the same bits (a + b = b + a), so half of the row is
// free for the partial sums of a second register.  DPP writes only the lanes its bank mask names (banks = groups of
// four lanes of a row), the others keep the `old` operand:
//   fold8(p, q)   lanes 0-7 of every row: p[l] + p[l + 8]        lanes 8-15: q[l] + q[l - 8]
//   fold4(r, s)   lanes with bit 2 clear: r[l] + r[l + 4]        lanes with bit 2 set: s[l] + s[l - 4]
// The pairs added are the pairs row_sum16 adds at these levels (l with l ^ 8, then l with l ^ 4 inside a half that is
// 8-periodic), so every partial sum has the bits it had; 16 registers cost 102 vector instructions instead of 192.
template <int CTRL, int BANKS>
__device__ __forceinline__ double lane_dpp_into(double old, double src) {   // old with the lanes of BANKS replaced by the rotated src
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xF, BANKS, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xF, BANKS, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double fold8(double p, double q) {
  const double a = lane_dpp_into<0x128, 0xC>(p, q);   // l < 8: p[l]          l >= 8: q[l - 8]
  const double b = lane_dpp_into<0x128, 0x3>(q, p);   // l < 8: p[l + 8]      l >= 8: q[l]
  return a + b;
}
__device__ __forceinline__ double fold4(double r, double s) {
  // row_ror:n moves data n lanes up (dst[l] = src[(l - n) mod 16]): s[l - 4] is ror 4, r[l + 4] is ror 12
  const double a = lane_dpp_into<0x124, 0xA>(r, s);   // bit 2 clear: r[l]        bit 2 set: s[l - 4]
  const double b = lane_dpp_into<0x12C, 0x5>(s, r);   // bit 2 clear: r[l + 4]    bit 2 set: s[l]
  return a + b;
}
// the last two levels, inside each bank (the neighbouring banks now belong to other registers): l with l ^ 2, then
// l with l ^ 1 — the operands row_sum16's rotations by 2 and by 1 meet, since the level-4 sums are 4-periodic
template <int CTRL>
__device__ __forceinline__ double lane_quad(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_sum4(double v) {
  v = v + lane_quad<0x4E>(v);   // quad_perm [2, 3, 0, 1]
  v = v + lane_quad<0xB1>(v);   // quad_perm [1, 0, 3, 2]
  return v;
}

// geom_scalars (ia3_model.h) across the lanes of the wave.  Its ~20 exponentials, ~20 divisions and 2 square roots are
// independent library sequences of 20-35 dependent instructions each; a wave that owns its SIMD (every fit does: one
// wave per SIMD) sits through all of them one after the other — 6.5 k cycles of the 38 k an LM iteration takes
// (scripts/lone_fit.py).  Here every lane takes one of them: stage 1 one exponential per lane, stage 2 one division
// per lane (operands prepared per kind, results post-processed per kind), stage 3 the reciprocals of the widths and
// the two cosines; values travel through 64 doubles of LDS.  Each lane performs the operations of the serial code on
// the same operands, so the scalars are the ones geom_scalars returns.
#define IA3_LDS __attribute__((address_space(3)))
// (x, cfg and sc live in LDS: typed pointers make these ds_read / ds_write instead of flat accesses, which the
// not-inlined evaluation would otherwise get because its pointers travel through memory)
__device__ __forceinline__ void geom_scalars_wave(const IA3_LDS double* x, const IA3_LDS FitCfg& cfg, GeomScalars& q,
                                                  IA3_LDS double* sc) {
  // (opaque to the optimiser: the lane-range predicates below would otherwise be computed once per kernel, kept as 64-bit
  // masks in scalar registers the geometry leaves no room for, and come back from their spill lanes with two
  // v_readlane each at every use — a compare is one instruction)
  int ln = threadIdx.x & 63;
  asm volatile("" : "+v"(ln));
  const int variant = cfg.variant;
  const double delta = cfg.delta, min_ws = cfg.min_ws, max_ws = cfg.max_ws;
  // ---- stage 1: sc[i] = exp(arg_i) ------------------------------------------------------------------------------
  //  0 tp   1 pp   2-4 w1..3   5-7 centre (variant 1: numerator exp(-a))   8-10 variant 1: exp(-b)
  //  11-13 -|xp|,-|yp|,-|zp|   14-16 -|w1..3|   17 -|pp|/2   18 -|tp|/2   19 bk   20 clipped bk   21 h (natural_wave)
  if (ln < 22) {
    double arg;
    if (ln < 2) arg = x[9 - ln];
    else if (ln < 5) arg = x[3 + ln];
    else if (ln < 8) arg = variant == 1 ? -x[ln == 5 ? 2 : 3] : x[ln - 3];
    else if (ln < 11) arg = -x[ln - 6];
    else if (ln < 14) arg = -fabs(x[ln - 9]);
    else if (ln < 17) arg = -fabs(x[ln - 9]);
    else if (ln < 19) arg = -fabs(x[ln - 9]) / 2;          // 17: pp = x[8], 18: tp = x[9]
    else if (ln == 19) arg = x[0];
    else if (ln == 21) arg = x[1];
    else { const double bk = x[0]; arg = bk < -709.78 ? -709.78 : (bk > 709.78 ? 709.78 : bk); }
    sc[ln] = exp(arg);
  }
  __builtin_amdgcn_wave_barrier();
  // ---- stage 2: r = A / B, then the per-kind tail ---------------------------------------------------------------
  //  0,1 t,p (sig_sine)   2-4 ws (sig_ws)   5-7 centre   8-10 d centre   11-13 norm_w   14 np_   15 nt_
  if (ln < 16) {
    double A, B, v = 0.0;
    const double dws = max_ws - min_ws;
    if (ln < 2) { v = x[9 - ln]; A = 2.; B = 1. + sc[ln]; }
    else if (ln < 5) { v = x[3 + ln]; A = dws; B = 1. + sc[ln]; }
    else if (ln < 8) {
      if (variant == 1) { A = 2. * delta * sc[ln]; B = 1. + sc[ln + 3]; }
      else { v = x[ln - 3]; A = 2. * delta; B = 1. + sc[ln]; }
    }
    else if (ln < 11) { const double e = sc[ln + 3]; A = -delta * e; B = (1 + e) * (1 + e); }
    else if (ln < 14) {
      const double w = x[ln - 6], e = sc[ln + 3];
      const double d = w > 0 ? max_ws * e + min_ws : min_ws * e + max_ws;
      A = 0.5 * (max_ws - min_ws) * e; B = d * d;
    }
    else { const double e = sc[ln + 3]; A = e; B = 1 + e * e; }
    double r = A / B;
    if (ln < 2) { r = r - 1.; if (v >= IA3_LOGMAX) r = -1.; if (v <= -IA3_LOGMAX) r = 1.; }
    else if (ln < 5) { r = r + min_ws; if (v >= IA3_LOGMAX) r = min_ws; if (v <= -IA3_LOGMAX) r = dws + min_ws; }
    else if (ln < 8) {
      const double c0 = cfg.c0[ln - 5];
      r = r - delta + c0;
      if (variant != 1) { if (v >= IA3_LOGMAX) r = -delta + c0; if (v <= -IA3_LOGMAX) r = delta + c0; }
    }
    sc[32 + ln] = r;
  }
  __builtin_amdgcn_wave_barrier();
  // ---- stage 3: 1 / ws_k (lanes 0-2), cosines (lanes 3, 4), widths sqrt(ws_k) (lanes 5-7, for natural_wave) -------
  if (ln < 3) sc[48 + ln] = 1. / sc[34 + ln];
  else if (ln < 8) {
    const double u = sc[29 + ln];                       // 3: t, 4: p, 5-7: ws_k
    const double u2 = u * u;
    sc[48 + ln] = sqrt(ln < 5 ? 1 - u2 : u);
  }
  __builtin_amdgcn_wave_barrier();
  q.t = sc[32]; q.p = sc[33];
  q.tc = sc[51]; q.pc = sc[52];
#pragma unroll
  for (int k = 0; k < 3; ++k) { q.s[k] = sc[48 + k]; q.c[k] = sc[37 + k]; q.nc[k] = sc[40 + k]; q.nw[k] = sc[43 + k]; }
  q.np_ = sc[46]; q.nt_ = sc[47];
  q.ebk_j = sc[19];
  q.ebk_f = variant == 1 ? sc[19] : sc[20];
  __builtin_amdgcn_wave_barrier();   // sc is reused by the next evaluation only after these reads
}

// to_natural (ia3_model.h) from what geom_scalars_wave left in sc: the same operations on the same operands as the
// serial form (whose dozen exponentials, divisions and square roots a lone wave sits through one after the other)
__device__ __forceinline__ void natural_wave(const IA3_LDS double* sc, IA3_LDS float* p) {
  const int ln = threadIdx.x & 63;
  if (ln == 0) {
    p[0] = (float)sc[21];                                                     // exp(h)
    p[1] = (float)sc[37]; p[2] = (float)sc[38]; p[3] = (float)sc[39];         // centre
    p[4] = (float)sc[19];                                                     // exp(bk)
    p[5] = (float)sc[53]; p[6] = (float)sc[54]; p[7] = (float)sc[55];         // sqrt(ws_k)
    p[8] = (float)sc[32]; p[9] = (float)sc[33];                               // sin t, sin p
  }
}

// Everything a fit keeps between its phases lives in LDS, one block per wave: a wave may hold no more than 256
// registers (two waves per SIMD), and the only phase that needs most of them is the evaluation (66 float64 sums per lane).
struct BallLds { float dat[SLOTS][64], cz[SLOTS][64], cx[SLOTS][64], cy[SLOTS][64]; };   // [slot][lane]: 8 KB per wave
struct WaveLds {
  BallLds bl;            // float32 data the fit sees (GaussianFit casts to float32, :172) and voxel coordinates
  LMWork w;
  FitCfg cfg;
  double gsc[64];        // geom_scalars_wave
  double lo10[10], hi10[10];   // the ten smallest / largest voxel values, each ascending (start point, :175-182)
  float p[12];           // the fit's row
  float co[4];           // centre of the previous fit of the seed (convergence test)
  // tallies of this wave, flushed with one atomic each when the wave leaves the kernel: the per-fit atomics on three
  // shared cache lines (counters, stage control, n_iter) were the kernel's largest wait at two waves per SIMD — 43 us
  // per work-list position behind the ticket draw, which queues behind them (profiles/r03a/fit_stamps.log)
  unsigned long long tally[3];   // fits run, function evaluations, voxel evaluations — of field tally_fov
  unsigned long long tally_wait; // shader cycles spent in dependency waits (refit admission)
  int tally_conv, tally_iter;    // seeds that converged; highest sweep made in field tally_fov
  int tally_fov;                 // the field the per-field tallies belong to (-1: none yet); flushed when it changes
#ifdef IA3_FIT_STAMPS
  unsigned long long t_last, stamp[24];   // profiling build only (scripts/fit_stamps.sh): shader cycles per phase
  unsigned long long stamp_fit[24];       // ... of the fit in progress (IA3_FIT_STAMPS == 2 keeps only fits beyond 100 evaluations)
#endif
};

// Profiling build (-DIA3_FIT_STAMPS, never the shipped library): cycles since the previous stamp are added to phase k.
#ifdef IA3_FIT_STAMPS
#define IA3_STAMP(L, k)                                                                       \
  do {                                                                                        \
    const unsigned long long t_ = __builtin_readcyclecounter();                               \
    if ((threadIdx.x & 63) == 0) {                                                            \
      if ((k) == 3 || (k) == 4 || ((k) >= 6 && (k) <= 10)) (L)->stamp_fit[k] += t_ - (L)->t_last;   \
      else (L)->stamp[k] += t_ - (L)->t_last;                                                 \
      (L)->t_last = t_;                                                                       \
    }                                                                                         \
    __builtin_amdgcn_wave_barrier();                                                          \
  } while (0)
// end of a fit: its phases join the wave's totals (all fits, or only the stragglers)
#define IA3_STAMP_FIT_END(L, nfev_)                                                           \
  do {                                                                                        \
    if ((threadIdx.x & 63) < 24) {                                                            \
      const int l_ = threadIdx.x & 63;                                                        \
      if (IA3_FIT_STAMPS != 2 || (nfev_) > 100) {                                             \
        (L)->stamp[l_] += (L)->stamp_fit[l_];                                                 \
        if (l_ == 15) (L)->stamp[15] += (unsigned long long)(nfev_);                          \
        if (l_ == 16) (L)->stamp[16] += 1ull;                                                 \
      }                                                                                       \
      (L)->stamp_fit[l_] = 0ull;                                                              \
    }                                                                                         \
    __builtin_amdgcn_wave_barrier();                                                          \
  } while (0)
#else
#define IA3_STAMP(L, k) do { } while (0)
#define IA3_STAMP_FIT_END(L, n) do { } while (0)
#endif

// Wave-parallel evaluation of |f|, JᵀJ, Jᵀf for lm_solve.
struct WaveEval {
  IA3_LDS WaveLds* L;
  unsigned valid;     // bit s: slot s of this lane holds a voxel
  __device__ __forceinline__ double eval(const double* x, double* A, double* g) {
    IA3_STAMP(L, 9);   // algebra since the last evaluation (or the fit's set-up before the first)
    Geom gm;   // wave-uniform coefficient tables, in scalar registers: the slot loop below reads them as SGPR operands
    const IA3_LDS BallLds* bl = &L->bl;
    {
      Geom g0;
      {
        GeomScalars gs;
        geom_scalars_wave((const IA3_LDS double*)x, L->cfg, gs, L->gsc);
        const double xh[2] = {0.0, ((const IA3_LDS double*)x)[1]};
        geom_assemble(xh, gs, g0);
      }
      gm.h = sgpr(g0.h); gm.ebk_f = sgpr(g0.ebk_f); gm.ebk_j = sgpr(g0.ebk_j);
#pragma unroll
      for (int k = 0; k < 3; ++k) gm.c[k] = sgpr(g0.c[k]);
#pragma unroll
      for (int k = 0; k < 6; ++k) gm.q[k] = sgpr(g0.q[k]);
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int a = 0; a < 3; ++a) gm.l[k][a] = sgpr(g0.l[k][a]);
#pragma unroll
      for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int a = 0; a < 6; ++a) gm.m[k][a] = sgpr(g0.m[k][a]);
    }
    IA3_STAMP(L, 6);   // (geometry; the time since the previous stamp up to the call is the solver's algebra)
    double a[NTRI], gg[NP], ss = 0.0;
#pragma unroll
    for (int k = 0; k < NTRI; ++k) a[k] = 0.0;
#pragma unroll
    for (int k = 0; k < NP; ++k) gg[k] = 0.0;
    const int ln = threadIdx.x & 63;
    // a rolled loop: one slot's exp -> Jacobian row -> 66 accumulations is a dependent chain that the SIMD's other
    // wave fills; unrolled, the allocator wants the whole register file
#pragma unroll 1
    for (int s = 0; s < SLOTS; ++s) {
      if (valid & (1u << s)) {
        double J[NP];
        double F = model_jac(gm, (double)bl->cz[s][ln], (double)bl->cx[s][ln], (double)bl->cy[s][ln], J);
        double r = (gm.ebk_f + F) - (double)bl->dat[s][ln];
        ss += r * r;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          gg[i] += J[i] * r;
#pragma unroll
          for (int j = i; j < NP; ++j) a[tri(i, j)] += J[i] * J[j];
        }
      }
    }
    IA3_STAMP(L, 7);   // voxel slots
    // sum the 65 partials over the wave: 65 -> 34 -> 17 registers by pairwise lane swaps, then inside the rows
    {
      constexpr int NV = NTRI + NP;            // 65
      constexpr int N1 = (NV + 1) / 2;         // 33 registers after the xor-32 level
      constexpr int N2 = (N1 + 1) / 2;         // 17 after the xor-16 level, four values each
      double p1[N1 + 1];
#pragma unroll
      for (int m = 0; m < N1; ++m) {
        const double x = 2 * m < NTRI ? a[2 * m] : gg[2 * m - NTRI];
        const double y = 2 * m + 1 < NTRI ? a[2 * m + 1] : (2 * m + 1 < NV ? gg[2 * m + 1 - NTRI] : 0.0);
        p1[m] = swap32_add(x, y);
      }
      p1[N1] = 0.0;
      const int lane = ln, row = lane >> 4;
      // row r of register n holds value 4n + {0, 2, 1, 3}[r]
      const int sel = row == 0 ? 0 : (row == 1 ? 2 : (row == 2 ? 1 : 3));
      static_assert(N2 == 17, "16 registers folded four to one + one on its own");
      double q2[N2];
#pragma unroll
      for (int n = 0; n < N2; ++n) q2[n] = swap16_add(p1[2 * n], p1[2 * n + 1]);
      // g follows A in the work area (LMWork): value vi lives at A + vi for every vi < NV
      IA3_LDS double* dst = (IA3_LDS double*)A;
      // inside the rows: registers 4k .. 4k+3 folded into one (fold8, fold4), whose lane l then holds the partial sum of
      // register 4k + 2 * bit2(l) + bit3(l); after the last two levels the four lanes of a bank agree
      const int vb = 4 * (2 * ((lane >> 2) & 1) + ((lane >> 3) & 1)) + sel;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double tot = row_sum4(fold4(fold8(q2[4 * k], q2[4 * k + 1]), fold8(q2[4 * k + 2], q2[4 * k + 3])));
        if ((lane & 3) == 0) dst[16 * k + vb] = tot;      // 16 k + vb <= 63 < NV
      }
      {
        const double tot = row_sum16(q2[16]);
        if ((lane & 15) == 0 && 64 + sel < NV) dst[64 + sel] = tot;
      }
      (void)g;
      __builtin_amdgcn_wave_barrier();   // A, g live in LDS: later reads by every lane follow these writes in order
    }
    // MINPACK's enorm (scaled sums) returns NaN, not inf, as soon as two components are infinite (inf/inf) or one
    // is NaN; lmder's tests `0.1*fnorm1 < fnorm` and `0.1*fnorm1 >= fnorm` are then both false, which changes
    // the trust-region update.  Reached by the legacy model (no clip on the background exponent) when a trial
    // step sends bk past 709.
    const double fn = sqrt(wave_sum(ss));
    IA3_STAMP(L, 8);   // cross-lane sums
    // A non-finite residual makes the sum of squares non-finite, so the count is only taken when that happened (the
    // residuals are evaluated again, by the same code; an overflow of the squares alone counts nothing)
    int nbad = 0;   // non-finite residuals, NaN counted twice
    if (!(fn - fn == 0.0)) {
#pragma unroll 1
      for (int s = 0; s < SLOTS; ++s) {
        bool r_inf = false, r_nan = false;
        if (valid & (1u << s)) {
          double J[NP];
          const double F = model_jac(gm, (double)bl->cz[s][ln], (double)bl->cx[s][ln], (double)bl->cy[s][ln], J);
          const double r = (gm.ebk_f + F) - (double)bl->dat[s][ln];
          r_nan = r != r;
          r_inf = !r_nan && (r - r != 0.0);
        }
        nbad += __popcll(__ballot(r_inf)) + 2 * __popcll(__ballot(r_nan));
      }
    }
    return nbad >= 2 ? NAN : fn;
  }
};

// The ten smallest and ten largest of the wave's valid values (each ascending) -> L->lo10, L->hi10.
// Every lane sorts its eight values once (19 compare-exchanges, empty slots = +inf); then ten rounds: wave minimum of
// the lanes' heads, the first lane that holds it drops its head.  The largest values are the smallest of the negated
// ones.  (The first form rescanned all eight slots of every lane in every round: 6.5 k instructions per ball.)
__device__ __forceinline__ void lane_sort8(double* v) {
#define IA3_CE(i, j) { const double lo_ = fmin(v[i], v[j]), hi_ = fmax(v[i], v[j]); v[i] = lo_; v[j] = hi_; }
  IA3_CE(0, 1) IA3_CE(2, 3) IA3_CE(4, 5) IA3_CE(6, 7)
  IA3_CE(0, 2) IA3_CE(1, 3) IA3_CE(4, 6) IA3_CE(5, 7)
  IA3_CE(1, 2) IA3_CE(5, 6) IA3_CE(0, 4) IA3_CE(3, 7)
  IA3_CE(1, 5) IA3_CE(2, 6)
  IA3_CE(1, 4) IA3_CE(3, 6)
  IA3_CE(2, 4) IA3_CE(3, 5)
  IA3_CE(3, 4)
#undef IA3_CE
}
struct OpFmin { __device__ __forceinline__ double operator()(double x, double y) const { return fmin(x, y); } };
template <bool NEG>
__device__ __forceinline__ void wave_smallest10(const double* vals, unsigned valid, IA3_LDS double* out) {
  const int lane = threadIdx.x & 63;
  double v[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) v[s] = (valid & (1u << s)) ? (NEG ? -vals[s] : vals[s]) : INFINITY;
  lane_sort8(v);
#pragma unroll 1
  for (int k = 0; k < 10; ++k) {
    const double gmin = wave_reduce(v[0], OpFmin());
    const unsigned long long m = __ballot(v[0] == gmin);
    if (lane == __ffsll((long long)m) - 1) {
#pragma unroll
      for (int s = 0; s + 1 < SLOTS; ++s) v[s] = v[s + 1];
      v[SLOTS - 1] = INFINITY;
    }
    if (lane == 0) out[NEG ? 9 - k : k] = NEG ? -gmin : gmin;
  }
}
__device__ __forceinline__ void wave_extremes(const double* vals, unsigned valid, IA3_LDS WaveLds* L) {
  wave_smallest10<false>(vals, valid, L->lo10);
  wave_smallest10<true>(vals, valid, L->hi10);
  __builtin_amdgcn_wave_barrier();
}

// One GaussianFit(...).fit() on the ball parked in L->bl (n >= 10 checked by the caller); start point from L->lo10 /
// L->hi10 (the float64 data before the float32 cast, :175-182).  The row goes to L->p.  centre_only: the caller needs
// nothing but the fitted centre p[1..3] (the first of two fits of a seed without neighbours): natural parameters and
// eps are left out.
__device__ __forceinline__ int wave_gaussfit(const FitArgs& fa, IA3_LDS WaveLds* L, unsigned valid, int kind, const double* c0,
                                             double delta, int n, bool centre_only) {
  IA3_LDS FitCfg& cfg_sh = L->cfg;   // wave-uniform: every lane writes the same values
  cfg_sh.min_ws = fa.min_ws; cfg_sh.max_ws = fa.max_ws; cfg_sh.delta = delta; cfg_sh.init_w = fa.init_w;
  cfg_sh.c0[0] = c0[0]; cfg_sh.c0[1] = c0[1]; cfg_sh.c0[2] = c0[2];
  cfg_sh.variant = fa.variant; cfg_sh.iw[0] = fa.iw[0]; cfg_sh.iw[1] = fa.iw[1]; cfg_sh.iw[2] = fa.iw[2];
  __builtin_amdgcn_wave_barrier();
  const FitCfg& cfg = *(const FitCfg*)&cfg_sh;
  LMWork& w = *(LMWork*)&L->w;
  float* p_out = (float*)L->p;
  WaveEval ev;
  ev.L = L;
  ev.valid = valid;
  {
    double lo10[10], hi10[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) { lo10[k] = L->lo10[k]; hi10[k] = L->hi10[k]; }
    init_guess(lo10, hi10, kind, cfg, w.x);
  }
  __builtin_amdgcn_wave_barrier();
  IA3_STAMP(L, 3);   // fit set-up, start point
  LMResult r = lm_solve(ev, w, fa.ftol, fa.xtol, fa.gtol, fa.maxfev, fa.factor);
  IA3_STAMP(L, 4);   // solver tail after the last evaluation
  // natural parameters and eps: the transcendental part once more through the lanes (geom_scalars_wave), at w.x
  Geom gm;
  {
    GeomScalars gs;
    // (the values stay in L->gsc after the call: nothing writes there before the next evaluation)
    geom_scalars_wave((const IA3_LDS double*)L->w.x, L->cfg, gs, L->gsc);
    if (centre_only) {
      if ((threadIdx.x & 63) == 0) { L->p[1] = (float)gs.c[0]; L->p[2] = (float)gs.c[1]; L->p[3] = (float)gs.c[2]; }
      __builtin_amdgcn_wave_barrier();
      IA3_STAMP(L, 10);   // natural parameters, eps
      IA3_STAMP_FIT_END(L, r.nfev);
      return r.nfev;
    }
    natural_wave(L->gsc, L->p);
    const double xh[2] = {0.0, L->w.x[1]};
    geom_assemble(xh, gs, gm);
  }
  double s = 0.0;
  const int ln = threadIdx.x & 63;
#pragma unroll
  for (int sl = 0; sl < SLOTS; ++sl)
    if (valid & (1u << sl))
      s += fabs((gm.ebk_f + model_f0(gm, (double)L->bl.cz[sl][ln], (double)L->bl.cx[sl][ln], (double)L->bl.cy[sl][ln])) -
                (double)L->bl.dat[sl][ln]);
  const double eps = wave_sum(s) / (double)n;
  if (ln == 0) L->p[10] = (float)eps;
  __builtin_amdgcn_wave_barrier();
  IA3_STAMP(L, 10);
  IA3_STAMP_FIT_END(L, r.nfev);
  return r.nfev;
}

// per-field tallies of the wave -> global memory (one atomic each; called when the wave moves on to another field and
// when it leaves the kernel)
__device__ __forceinline__ void flush_tallies(const FitArgs& fa, IA3_LDS WaveLds* L) {
  if ((threadIdx.x & 63) == 0 && L->tally_fov >= 0) {
    if (L->tally[0]) {
      atomicAdd(&fa.counters[0], L->tally[0]); atomicAdd(&fa.counters[1], L->tally[1]); atomicAdd(&fa.counters[2], L->tally[2]);
      unsigned long long* fc = fa.fov_counters + 4 * (size_t)L->tally_fov;
      atomicAdd(&fc[0], L->tally[0]); atomicAdd(&fc[1], L->tally[1]); atomicAdd(&fc[2], L->tally[2]);
    }
    if (L->tally_iter) atomicMax(&fa.n_iter[L->tally_fov], L->tally_iter);
    L->tally[0] = L->tally[1] = L->tally[2] = 0ull;
    L->tally_iter = 0;
  }
}
__device__ __forceinline__ void tally_field(const FitArgs& fa, IA3_LDS WaveLds* L, int fov) {
  if (__builtin_amdgcn_readfirstlane(L->tally_fov) != fov) {
    flush_tallies(fa, L);
    if ((threadIdx.x & 63) == 0) L->tally_fov = fov;
    __builtin_amdgcn_wave_barrier();
  }
}

// Hand a fit's results over: the seed's state record (unconstrained parameters, delta, flags) and its row, in ONE store
// instruction — lane l < 26 writes dword l of the SeedState, lanes 32..42 the eleven floats of the row.  (Lane 0 writing
// 25 values one after the other cost 29 us per work-list position at two waves per SIMD: write-through `sc1` stores of
// one wave complete one at a time, ~1.2 us each under load, and the next ticket draw waits for all of them;
// profiles/r03a/fit_stamps.log.)  Readers take the record only after done[i] has been raised behind the drained store.
__device__ __forceinline__ void store_result(const FitArgs& fa, int i, IA3_LDS WaveLds* L,
                                             double delta, bool ok, int n, int nfev, bool write_conv, bool cv) {
  const int lane = threadIdx.x & 63;
  static_assert(sizeof(SeedState) == 26 * 4 && offsetof(SeedState, delta) == 80 && offsetof(SeedState, success) == 88 &&
                offsetof(SeedState, has_rec) == 92 && offsetof(SeedState, conv) == 96 && offsetof(SeedState, ver) == 100,
                "store_result writes SeedState by dwords");
  unsigned v = 0u;
  bool act = false;
  if (lane < 20) { v = ((const IA3_LDS unsigned*)L->w.x)[lane]; act = ok; }
  else if (lane < 22) { v = lane == 20 ? (unsigned)__double2loint(delta) : (unsigned)__double2hiint(delta); act = ok; }
  else if (lane == 22) { v = ok ? 1u : 0u; act = true; }
  else if (lane == 23) { v = 1u; act = ok; }
  else if (lane == 24) { v = cv ? 1u : 0u; act = write_conv; }
  else if (lane == 25) { v = (unsigned)LDH(&fa.state[i].ver) + 1u; act = ok; }   // only this wave ever writes the record
  else if (lane >= 32 && lane < 43) { v = ((const IA3_LDS unsigned*)L->p)[lane - 32]; act = ok; }
  unsigned* dst = lane < 32 ? (unsigned*)&fa.state[i] + lane : (unsigned*)&fa.ps[(size_t)i * 11] + (lane - 32);
  if (act) st_sc1(dst, v);
  if (lane == 0) {
    fa.nvox[i] = n;
    fa.nfev[i] += nfev;
    if (ok) {
      L->tally[0] += 1ull;
      L->tally[1] += (unsigned long long)nfev;
      L->tally[2] += (unsigned long long)nfev * (unsigned long long)n;
    }
  }
}

// Squared distance as scipy's cKDTree forms it (query.cxx: s = 0; s += d_k * d_k for k = 0, 1, 2; separate multiply and
// add): the Voronoi decisions compare these values for equality, so no contraction into fused multiply-adds here.
__device__ __forceinline__ double dist2_seq(double az, double ax, double ay, double bz, double bx, double by) {
  const double d0 = az - bz, d1 = ax - bx, d2 = ay - by;
  return __dadd_rn(__dadd_rn(__dmul_rn(d0, d0), __dmul_rn(d1, d1)), __dmul_rn(d2, d2));
}

// offsets of ball voxel vi from the truncated seed position (packed signed bytes dz, dx, dy)
struct BallOff { int dz, dx, dy; };
__device__ __forceinline__ BallOff ball_off(const int* __restrict__ ball, int vi) {
  const int w = ball[vi];
  return BallOff{(int)(signed char)(w & 0xff), (int)(signed char)((w >> 8) & 0xff), (int)(signed char)((w >> 16) & 0xff)};
}

struct StageCtl;
// ---- neighbour lists on the device: one wave per seed scans all seeds 64 at a time; hits are appended with
// ballot + prefix rank, so every list comes out in ascending index order (5 k seeds: 25 M distance tests, ~10 µs;
// the seed list never goes back to the host for this).  For every neighbour found the wave also looks for EXACT Voronoi
// ties — a voxel of seed i's ball, inside the image, as far from seed j as from seed i — and flags the seed and the
// fitter: the reference resolves those by cKDTree's traversal order (ia3_kdtree.h), which needs a tree built on the host.
__global__ __launch_bounds__(256) void nbr_build_k(const double* __restrict__ seeds, int n_all, double r2, int* __restrict__ cnt,
                                                   int* __restrict__ idx, int* __restrict__ overflow,
                                                   const int* __restrict__ ball, int nball, int Z, int X, int Y,
                                                   int* __restrict__ tie_flag, int* __restrict__ ctl_ties,
                                                   const int* __restrict__ fov_of, const int* __restrict__ fov_start,
                                                   int* __restrict__ fov_ties, unsigned long long* __restrict__ fov_with_nbr) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_all) return;   // whole wave leaves together
  const int fov = fov_of[i];
  const int first = fov_start[fov], n = fov_start[fov + 1];   // candidates: the seeds of the same field
  if (i >= n) return;   // a grid sized for a capacity (the count was still on the device at launch time)
  const double cz = seeds[3 * i], cx = seeds[3 * i + 1], cy = seeds[3 * i + 2];
  const int iz = (int)cz, ix = (int)cx, iy = (int)cy;
  int c = 0;
  bool tie = false;
  for (int j0 = first; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    bool hit = false;
    if (j < n && j != i) {
      const double a = cz - seeds[3 * j], b = cx - seeds[3 * j + 1], d = cy - seeds[3 * j + 2];
      hit = a * a + b * b + d * d <= r2;
    }
    unsigned long long m = __ballot(hit);
    if (hit) {
      const int pos = c + __popcll(m & ((1ull << lane) - 1ull));
      if (pos < MAXNB) idx[(size_t)i * MAXNB + pos] = j;
    }
    c += __popcll(m);
    while (m) {   // wave-uniform: the ball of seed i against neighbour j0 + b
      const int bj = __ffsll((long long)m) - 1;
      m &= m - 1;
      const double sz = seeds[3 * (j0 + bj)], sx = seeds[3 * (j0 + bj) + 1], sy = seeds[3 * (j0 + bj) + 2];
      for (int s = 0; s < SLOTS; ++s) {
        const int vi = lane + 64 * s;
        if (vi < nball) {
          const BallOff o = ball_off(ball, vi);
          const int z = iz + o.dz, x = ix + o.dx, y = iy + o.dy;
          if (z >= 0 && z < Z && x >= 0 && x < X && y >= 0 && y < Y)
            tie |= dist2_seq(cz, cx, cy, (double)z, (double)x, (double)y) == dist2_seq(sz, sx, sy, (double)z, (double)x, (double)y);
        }
      }
    }
  }
  const bool any_tie = __ballot(tie) != 0ull;
  if (lane == 0) {
    if (c > MAXNB) atomicMax(overflow, c);   // statistics only: consumers fall back to each_neighbour's scan
    cnt[i] = c;
    tie_flag[i] = any_tie ? 1 : 0;
    // (flags: a look before the atomic — a crowded field would otherwise send thousands of them to one word; the count of
    // seeds with neighbours goes to the seed's own field, fit_stages_k adds the fields up)
    if (any_tie) {
      if (!__hip_atomic_load(ctl_ties, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(ctl_ties, 1);
      if (!__hip_atomic_load(&fov_ties[fov], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&fov_ties[fov], 1);
    }
    if (c > 0) atomicAdd(&fov_with_nbr[4 * (size_t)fov + 3], 1ull);
  }
}

// f(j) for every seed j != i within 2r of seed i, ascending j, wave-uniform.  Seeds with at most MAXNB neighbours read
// their list; denser ones (the reference has no cap: Fitting_v4.py:601,612 query a cKDTree) scan the whole seed list,
// 64 candidates per step with a ballot.  f returns false to stop early.
template <class F>
__device__ __forceinline__ void each_neighbour(const FitArgs& fa, int i, F f) {
  const int cnt = fa.nbr_cnt[i];
  if (cnt <= fa.nb_cap) {
    for (int q = i * MAXNB; q < i * MAXNB + cnt; ++q)
      if (!f(fa.nbr_idx[q])) return;
    return;
  }
  const int lane = threadIdx.x & 63;
  const double cz = fa.seeds[3 * i], cx = fa.seeds[3 * i + 1], cy = fa.seeds[3 * i + 2];
  const int fov = fa.fov_of[i];
  const int first = fa.fov_start[fov], last = fa.fov_start[fov + 1];   // neighbours live in the same field
  for (int j0 = first; j0 < last; j0 += 64) {
    const int j = j0 + lane;
    bool hit = false;
    if (j < last && j != i) {
      const double a = cz - fa.seeds[3 * j], b = cx - fa.seeds[3 * j + 1], d = cy - fa.seeds[3 * j + 2];
      hit = a * a + b * b + d * d <= fa.nb_r2;
    }
    unsigned long long m = __ballot(hit);
    while (m) {
      const int b = __ffsll((long long)m) - 1;
      m &= m - 1;
      if (!f(j0 + b)) return;
    }
  }
}

// ---- stage 0 = firstfit of one seed (Fitting_v4.py:606-637) -----------------------------------------
// the voxels of the seed's Voronoi cell inside its ball, from the ORIGINAL image; returns their number
__device__ __forceinline__ int gather_first(const FitArgs& fa, int i, IA3_LDS BallLds* bl, unsigned& valid_out, double* vals) {
  const int lane = threadIdx.x & 63;
  const double c0[3] = {fa.seeds[3 * i], fa.seeds[3 * i + 1], fa.seeds[3 * i + 2]};
  const int iz = (int)c0[0], ix = (int)c0[1], iy = (int)c0[2];  // Python int(): toward zero
  // Voronoi (:612, :422-424): a voxel is dropped if another seed is strictly nearer, or equally near and the
  // reference's cKDTree query meets that one first — a property of the tree, resolved beforehand into fa.tie_lost
  // (voronoi_ties_k).  Without masks (nbr_build_k found no tie in the whole field) the tie branch is never taken.
  unsigned lost = 0;   // bit s = slot s belongs to another seed's Voronoi cell
  if (fa.nbr_cnt[i] > 0) {
    unsigned tied = 0;
    each_neighbour(fa, i, [&](int j) {
      const double sz = fa.seeds[3 * j], sx = fa.seeds[3 * j + 1], sy = fa.seeds[3 * j + 2];
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        const int vi = lane + 64 * s;
        if (vi < fa.nball) {
          const BallOff o = ball_off(fa.ball, vi);
          const double z = (double)(iz + o.dz), x = (double)(ix + o.dx), y = (double)(iy + o.dy);
          const double dme = dist2_seq(c0[0], c0[1], c0[2], z, x, y);
          const double dj = dist2_seq(sz, sx, sy, z, x, y);
          if (dj < dme) lost |= 1u << s;
          else if (dj == dme) { tied |= 1u << s; if (!fa.tie_lost && j < i) lost |= 1u << s; }
        }
      }
      return true;
    });
    if (fa.tie_lost && __ballot(tied != 0u)) {
#pragma unroll
      for (int s = 0; s < SLOTS; ++s)
        if ((fa.tie_lost[(size_t)i * SLOTS + s] >> lane) & 1ull) lost |= tied & (1u << s);
    }
  }
  unsigned valid = 0;
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int vi = lane + 64 * s;
    float fd = 0.f, fz = 0.f, fx = 0.f, fy = 0.f;
    vals[s] = 0.0;
    if (vi < fa.nball) {
      const BallOff o = ball_off(fa.ball, vi);
      const int z = iz + o.dz, x = ix + o.dx, y = iy + o.dy;
      const bool ok = z >= 0 && z < fa.Z && x >= 0 && x < fa.X && y >= 0 && y < fa.Y && !(lost & (1u << s));
      if (ok) {
        const double v = load_voxel(fa.ims[fa.fov_of[i]], fa.dtype, ((size_t)z * fa.X + x) * fa.Y + y);
        valid |= 1u << s;
        fd = (float)v; vals[s] = v;
        fz = (float)z; fx = (float)x; fy = (float)y;
      }
    }
    bl->dat[s][lane] = fd; bl->cz[s][lane] = fz; bl->cx[s][lane] = fx; bl->cy[s][lane] = fy;
  }
  valid_out = valid;
  return __popcll(__ballot(valid & 1u)) + __popcll(__ballot(valid & 2u)) + __popcll(__ballot(valid & 4u)) +
         __popcll(__ballot(valid & 8u)) + __popcll(__ballot(valid & 16u)) + __popcll(__ballot(valid & 32u)) +
         __popcll(__ballot(valid & 64u)) + __popcll(__ballot(valid & 128u));
}

// ---- exact Voronoi ties by the reference's rule (ia3_kdtree.h) -----------------------------------------------------
// One wave per seed that nbr_build_k flagged.  Every in-image voxel of the seed's ball that is exactly as far from the
// seed as from its nearest other seed asks the tree (one query per lane, the lane's queue in LDS): the voxel stays with
// the seed iff cKDTree.query(voxel) returns the seed.  Result: per slot a lane mask of lost tie voxels.
constexpr int KDQ_CAP = 24;   // queue entries per lane (realistic fields need <= 5, a 4 600-seed blob of 6 px sigma 16)
// Launched per field of view: `tree` is built from that field's seeds (its point indices are local: seed - seed0).
__global__ __launch_bounds__(64) void voronoi_ties_k(FitArgs fa, KdTree tree, int seed0, const int* __restrict__ tie_flag,
                                                     unsigned long long* __restrict__ tie_lost, int* __restrict__ ctl_abort,
                                                     int qcap) {
  __shared__ KdQEntry heap[KDQ_CAP][64];
  const int i = seed0 + (int)blockIdx.x;
  if (i >= seed0 + tree.n || !tie_flag[i]) return;
  const int lane = threadIdx.x & 63;
  const double c0[3] = {fa.seeds[3 * i], fa.seeds[3 * i + 1], fa.seeds[3 * i + 2]};
  const int iz = (int)c0[0], ix = (int)c0[1], iy = (int)c0[2];
  unsigned lost = 0, tied = 0;
  each_neighbour(fa, i, [&](int j) {
    const double sz = fa.seeds[3 * j], sx = fa.seeds[3 * j + 1], sy = fa.seeds[3 * j + 2];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int vi = lane + 64 * s;
      if (vi < fa.nball) {
        const BallOff o = ball_off(fa.ball, vi);
          const double z = (double)(iz + o.dz), x = (double)(ix + o.dx), y = (double)(iy + o.dy);
        const double dme = dist2_seq(c0[0], c0[1], c0[2], z, x, y);
        const double dj = dist2_seq(sz, sx, sy, z, x, y);
        if (dj < dme) lost |= 1u << s;
        else if (dj == dme) tied |= 1u << s;
      }
    }
    return true;
  });
  tied &= ~lost;   // a strictly nearer seed decides without the tree
  bool overflow = false;
#pragma unroll 1
  for (int s = 0; s < SLOTS; ++s) {
    bool lose = false;
    const int vi = lane + 64 * s;
    if ((tied >> s) & 1u) {
      const BallOff o = ball_off(fa.ball, vi);
      const int z = iz + o.dz, x = ix + o.dx, y = iy + o.dy;
      if (z >= 0 && z < fa.Z && x >= 0 && x < fa.X && y >= 0 && y < fa.Y) {
        const double q[3] = {(double)z, (double)x, (double)y};
        KdQueue<IA3_LDS KdQEntry*> queue((IA3_LDS KdQEntry*)&heap[0][lane], 64, qcap);
        const int w = kd_nearest(tree, q, 2.0 * fa.radius, queue);
        overflow |= queue.overflow;
        lose = w != i - seed0;
      }
    }
    const unsigned long long m = __ballot(lose);
    if (lane == 0) tie_lost[(size_t)i * SLOTS + s] = m;
  }
  // a query that ran out of queue entries: the host repeats the tie queries with the same tree and an unbounded queue
  // (resolve_ties_host) before the fit is launched again
  if (__ballot(overflow) && lane == 0) atomicMax(ctl_abort, 3);
}

// ---- stage k >= 1 = one seed's refit in sweep k of repeatfit (:651-680) ---------------------------------
// the full ball, data = image minus the current reconstructions of the overlapping seeds; returns the voxel count
__device__ __forceinline__ int gather_repeat(const FitArgs& fa, int i, IA3_LDS BallLds* bl, unsigned& valid_out, double* vals) {
  const int lane = threadIdx.x & 63;
  const int r = fa.radius;
  const double c0[3] = {fa.seeds[3 * i], fa.seeds[3 * i + 1], fa.seeds[3 * i + 2]};
  const int iz = (int)c0[0], ix = (int)c0[1], iy = (int)c0[2];
  unsigned valid = 0;
  int vz[SLOTS], vx[SLOTS], vy[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int vi = lane + 64 * s;
    vz[s] = 0; vx[s] = 0; vy[s] = 0; vals[s] = 0.0;
    if (vi < fa.nball) {
      const BallOff o = ball_off(fa.ball, vi);
      const int z = iz + o.dz, x = ix + o.dx, y = iy + o.dy;
      if (z >= 0 && z < fa.Z && x >= 0 && x < fa.X && y >= 0 && y < fa.Y) {
        valid |= 1u << s;
        vals[s] = load_voxel(fa.ims[fa.fov_of[i]], fa.dtype, ((size_t)z * fa.X + x) * fa.Y + y);
        vz[s] = z; vx[s] = x; vy[s] = y;
      }
    }
  }
  // subtract the current reconstructions of the overlapping seeds (= im_add + own rec, :658-662)
  each_neighbour(fa, i, [&](int j) {
    const SeedState& sj = fa.state[j];
    if (!LDH(&sj.has_rec)) return true;
    FitCfg cj;
    cj.min_ws = fa.min_ws; cj.max_ws = fa.max_ws; cj.delta = LDH(&sj.delta); cj.init_w = fa.init_w;
    cj.variant = fa.variant;
    cj.c0[0] = fa.seeds[3 * j]; cj.c0[1] = fa.seeds[3 * j + 1]; cj.c0[2] = fa.seeds[3 * j + 2];
    const int jz = (int)cj.c0[0], jx = (int)cj.c0[1], jy = (int)cj.c0[2];
    double xj[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) xj[k] = LDH(&sj.x[k]);
    Geom gj;
    make_geom(xj, cj, gj);
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      if (valid & (1u << s)) {
        const int oz = vz[s] - jz, ox = vx[s] - jx, oy = vy[s] - jy;
        if (oz >= -r && oz < r && ox >= -r && ox < r && oy >= -r && oy < r && oz * oz + ox * ox + oy * oy <= r * r)
          vals[s] -= model_f0(gj, (double)vz[s], (double)vx[s], (double)vy[s]);
      }
    }
    return true;
  });
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    bl->dat[s][lane] = (float)vals[s];
    bl->cz[s][lane] = (float)vz[s]; bl->cx[s][lane] = (float)vx[s]; bl->cy[s][lane] = (float)vy[s];
  }
  valid_out = valid;
  return __popcll(__ballot(valid & 1u)) + __popcll(__ballot(valid & 2u)) + __popcll(__ballot(valid & 4u)) +
         __popcll(__ballot(valid & 8u)) + __popcll(__ballot(valid & 16u)) + __popcll(__ballot(valid & 32u)) +
         __popcll(__ballot(valid & 64u)) + __popcll(__ballot(valid & 128u));
}

// ---- one work-list position ------------------------------------------------------------------------------------------
// mode 0: first fit (stage 0).  mode 1: refit in sweep k >= 1; returns "converged" (:677-680).
// mode 2: a seed WITHOUT neighbours, stage 0 and sweep 1 by one wave.  No other seed's ball overlaps this one, so its
//   Voronoi cell is the whole ball and the residual image it refits in sweep 1 is the original image: both fits see the
//   SAME voxels and values (they differ in delta_center and in the arithmetic of the start point, Fitting_v4.py:175-182 on
//   float32 / integer data vs on the float64 residual).  The wave gathers the ball once, fits twice and hands over once;
//   the work-list position of the seed's sweep 1 finds done[i] >= 2 and leaves.  Same operations on the same operands
//   as the two separate positions; returns "converged".
// The fit itself has ONE call site (the kernel is instruction-cache bound enough as it is).
__device__ __forceinline__ bool run_position(const FitArgs& fa, IA3_LDS WaveLds* L, int i, int mode) {
  tally_field(fa, L, fa.fov_of[i]);
  const double c0[3] = {fa.seeds[3 * i], fa.seeds[3 * i + 1], fa.seeds[3 * i + 2]};
  unsigned valid = 0;
  int n;
  {
    double vals[SLOTS];
    n = mode == 1 ? gather_repeat(fa, i, &L->bl, valid, vals) : gather_first(fa, i, &L->bl, valid, vals);
    IA3_STAMP(L, 1);   // gather
    if (n >= NP) wave_extremes(vals, valid, L);   // mode 2: the same ten smallest / largest values start both fits
    IA3_STAMP(L, 2);   // extremes
  }
  int success_old = 0;
  float co0 = 0.f, co1 = 0.f, co2 = 0.f;
  if (mode == 1) {
    success_old = LDH(&fa.state[i].success);
    co0 = LDH(&fa.ps[(size_t)i * 11 + 1]); co1 = LDH(&fa.ps[(size_t)i * 11 + 2]); co2 = LDH(&fa.ps[(size_t)i * 11 + 3]);
  }
  const bool ok = n >= NP;  // :382-383 (mode 2: for both fits, same voxels)
  int nfev = 0, nfev_first = 0;
  const int npass = mode == 2 ? 2 : 1;
#pragma unroll 1
  for (int pass = 0; pass < npass; ++pass) {
    const bool refit = mode == 1 || pass == 1;
    if (pass == 1) { success_old = ok ? 1 : 0; co0 = L->p[1]; co1 = L->p[2]; co2 = L->p[3]; nfev_first = nfev; }
    // a refit sees the float64 residual (kind 2) and casts it to float32 (:172); the first fit the stack's own dtype
    if (ok) nfev = wave_gaussfit(fa, L, valid, refit ? 2 : (fa.dtype == IA3_F32 ? 0 : 1), c0,
                                 refit ? fa.delta_repeat : fa.delta_first, n, mode == 2 && pass == 0);
  }
  if (mode == 2 && ok && (threadIdx.x & 63) == 0) L->tally[0] += 1ull;   // two fits; store_result counts one
  // convergence (:677-680): float32 centre differences, compared in float64
  bool cv = true;
  if (mode != 0 && ok && success_old) {
    const float d0 = co0 - L->p[1], d1 = co1 - L->p[2], d2 = co2 - L->p[3];
    const float dist = (d0 * d0 + d1 * d1) + d2 * d2;
    cv = (double)dist < fa.dist_th2;
  }
  store_result(fa, i, L, mode == 0 ? fa.delta_first : fa.delta_repeat, ok, n, nfev + nfev_first, mode != 0, cv);
  IA3_STAMP(L, 11);   // store_result
  return cv;
}

// ---- dependency-ordered fit kernel ------------------------------------------------------------------------
// Work list = stages x seeds, stage-major: stage 0 is firstfit, stage k>=1 is sweep k of repeatfit; inside a
// stage the seeds come in index order.  Every block draws the next position
// with one atomic and waits until the fits it depends on have published:
//     (i, k>=1) needs   own stage k-1,   neighbours j < i at stage k   (already refitted in this sweep, the
//     reference's in-place Gauss–Seidel order),   neighbours j > i at stage k-1.
// done[j] = number of stages seed j has completed.  All dependencies of a position lie EARLIER in the list, so
// whoever holds the earliest unfinished position can always run: no deadlock, no co-residency requirement.
// Hand-off: producer sc1 stores -> vmcnt(0) -> relaxed agent store of done[j]; consumer relaxed agent poll of
// done[j] -> sc1 loads of the payload (MI355X_MICROARCH.md "Valid forms": every load and store of the handed-off
// words is sc1, the storing lane drains before it raises the counter, the polling wave loads only afterwards).
// Work is handed out through NCLAIM + 1 ticket counters, each on a cache line of its own: an agent-scope atomic on ONE
// word is served at ~50-80 ns apiece whoever asks (the XCDs' L2s are not coherent, the operation runs at the memory side),
// so the 12 000 draws of a 5 000-seed field from a single counter held every wave 27 us per draw at two waves per SIMD
// and bounded the kernel from below (profiles/r03a/fit_stamps.log).
//   claim[0 .. NCLAIM-1]  stage 0 (first fits; they wait for nothing): the seeds are split into NCLAIM ranges, a wave
//                         starts with the range of its block index and goes on to the others when that one is empty
//   claim[NCLAIM]         later stages, in UNITS of up to 64 consecutive positions (one lane looks at each: most are
//                         skips).  A unit is worked through by ONE wave, in order, so its size is chosen such that a
//                         unit is expected to hold less than one position that really needs a refit: 64 for a field
//                         of isolated spots, 1 for a crowded one (unit_size below)
// A wave draws from claim[NCLAIM] only after it has seen every stage-0 range exhausted, i.e. when every first fit is in the
// hands of a running wave; units are drawn in list order and worked through in order.  So whatever a position waits
// for is either a first fit (running or done) or an earlier position of an earlier-or-same unit (running or done): the
// argument that needs no co-residency stands.
constexpr int NCLAIM = 8;
struct StageCtl {
  int n_unconv;             // seeds not yet converged (repeat stages stop when it reaches 0)
  int abort;                // 1: a spin-wait exceeded its bound (never expected); 2: exact Voronoi ties exist and no tie masks
                            // were supplied (the host resolves them and launches again); 3: tie queue overflow
  int ties;                 // nbr_build_k: some ball voxel is equidistant from its seed and another one
  int stage_reached;        // the stage this launch ran up to (a launch from stage 0 decides that itself, see fit_stages_k)
  int pad[28];
  struct Claim { unsigned int next; unsigned int pad[31]; } claim[NCLAIM + 1];
};
static_assert(sizeof(StageCtl) == 128 * (NCLAIM + 2), "one 128-byte line per counter");

__device__ long long d_wait_bound = 1LL << 22;   // polls before a dependency wait gives up (IA3_DEBUG_FIT_WAITBOUND)

// wave-uniform poll: every lane issues the (same-address) load, lane 0's value decides for the whole wave
__device__ __forceinline__ bool wait_done(const int* done, int j, int need, StageCtl* ctl) {
  // An ordinary refit is over in tens of microseconds and its successor should notice at once; a wave that holds a later
  // sweep of a seed whose fit runs for milliseconds (thousands of them in a batch of uint16 fields) must not keep two
  // memory-side loads per 0.4 us going for all that time: the naps grow with the wait, the abort word — one address for
  // every waiting wave of the launch — is looked at every 16th time.
  long long spins = 0;
  for (;;) {
    const int v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&done[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (v >= need) { if (spins) __builtin_amdgcn_s_setprio(3); return true; }
    if (spins == 0) __builtin_amdgcn_s_setprio(0);   // a waiting wave takes no issue slot from the fits beside it
    if (spins < 256) __builtin_amdgcn_s_sleep(16);                                          // 0.4 us: the first 0.1 ms
    else if (spins < 1024) __builtin_amdgcn_s_sleep(64);                                    // 1.7 us: up to 1.4 ms
    else { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); }                  // 6.8 us
    ++spins;
    if ((spins & 15) == 0) {
      const int ab = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      if (spins > d_wait_bound || ab) {   // 2^22 polls = 28 s of waiting: never expected
        __hip_atomic_store(&ctl->abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
  }
}

__device__ __forceinline__ void publish(int* done, int i, int value) {
  if ((threadIdx.x & 63) == 0) {   // the same lane issued every hand-off store of this fit
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // max, not store: an early-exiting wave may already have marked this seed "all stages done" (1 << 20)
    // while the holder of an earlier position of the same seed publishes its smaller stage count afterwards
    __hip_atomic_fetch_max(&done[i], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// One block (= one wave) per work-list position.  The position is a TICKET drawn when the block starts running,
// not blockIdx: tickets are handed out in the order blocks actually start, so everything a block may wait for is
// held by a block that is already running (or done) whatever order the dispatcher picks.
// done[i] := max(done[i], value) without the release: for a seed whose data no wave of this kernel reads any more
__device__ __forceinline__ void publish_quiet(int* done, int i, int value) {
  if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_max(&done[i], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one work-list position; false = abort (a dependency wait exceeded its bound)
__device__ __forceinline__ bool stage_position(const FitArgs& fa, IA3_LDS WaveLds* L, int stage1, StageCtl* ctl,
                                               int* done, int k, int i) {
  const int lane = threadIdx.x & 63;
  // run_position (the fit) has ONE call site below: the kernel is instruction-cache bound, a second copy of the
  // solver costs more than the branches around this one.
  int mode;
  unsigned long long ver_sum = 0ull;
  if (k == 0) {
    // sweep 1 is part of this launch and nothing overlaps this seed: both of its fits from this wave (run_position)
    const bool fused = fa.fuse && stage1 >= 2 && fa.nbr_cnt[i] == 0;
    mode = fused ? 2 : 0;
  } else {
    // sweep k of this seed already made by the wave of an earlier position (the fused first fit)
    if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&done[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= k + 1) return true;
    // nothing left to refit anywhere: every remaining position is a skip.  The claimed position is still
    // published (as "all stages done") so that a block which passed this check earlier and waits on it can go on.
    if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&ctl->n_unconv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) <= 0) {
      publish(done, i, 1 << 20);
      return true;
    }
    const unsigned long long t_wait = __builtin_readcyclecounter();
    if (!wait_done(done, i, k, ctl)) return false;
    // ... or made meanwhile: the wave that holds this seed's first fit may have been running until now
    if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&done[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= k + 1) return true;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (__builtin_amdgcn_readfirstlane(LDH(&fa.state[i].conv))) { publish(done, i, k + 1); return true; }   // converged: skipped (:652)
    bool alive = true;
    each_neighbour(fa, i, [&](int j) { alive = wait_done(done, j, j < i ? k + 1 : k, ctl); return alive; });
    if (!alive) return false;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (lane == 0) L->tally_wait += __builtin_readcyclecounter() - t_wait;   // cycles this wave sat in dependency waits
    mode = 1;
    // What this refit would see of its neighbours (their records are final for this position: the waits above).  Unchanged
    // since this seed's previous refit = the same voxels, the same data, the same start point: the fit would return the
    // row it returned then, the centre would not move, the seed would be marked converged (:677-680).  That is what
    // happens here, without the fit — the last sweep of a plateau twin that refits noise to maxfev (15 ms for one wave
    // while the device idles: a lone uint16 FOV), of every seed whose neighbours had settled a sweep earlier.
    each_neighbour(fa, i, [&](int j) { ver_sum += (unsigned long long)(unsigned)LDH(&fa.state[j].ver); return true; });
    ver_sum |= 1ull << 63;
    if (k >= 2 && fa.memo && __builtin_amdgcn_readfirstlane((int)(LDH(&fa.memo[i]) == ver_sum))) {
      tally_field(fa, L, fa.fov_of[i]);
      if (lane == 0) {
        st_sc1(&fa.state[i].conv, 1);
        if (k > L->tally_iter) L->tally_iter = k;
        L->tally_conv += 1;
      }
      publish(done, i, k + 1);
      return true;
    }
  }
  IA3_STAMP(L, 13);   // admission (dependency waits)
  const bool cv = run_position(fa, L, i, mode);
  if (mode != 0 && lane == 0 && fa.memo) st_sc1(&fa.memo[i], mode == 2 ? (1ull << 63) : ver_sum);   // (mode 2: no neighbours)
  if (mode == 0) {
    publish(done, i, 1);
    return true;
  }
  if (lane == 0) {   // (the seed's conv flag went out with its record, store_result)
    const int sweep = mode == 2 ? 1 : k;
    if (sweep > L->tally_iter) L->tally_iter = sweep;
    if (cv) L->tally_conv += 1;
  }
  // fused, converged and without neighbours: no wave of this kernel will read this seed's rows or state again (its later
  // positions leave at the done[] test above), so the hand-over needs no release; the end of the kernel publishes it
  if (mode == 2 && cv) publish_quiet(done, i, 1 << 20);
  else publish(done, i, mode == 2 ? 2 : k + 1);
  IA3_STAMP(L, 12);   // hand-over
#ifdef IA3_FIT_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  IA3_STAMP(L, 14);   // drain of this position's stores (in the shipped kernel the next ticket draw waits for them)
#endif
  return true;
}

// One wave per block, at most two blocks per SIMD of the device (256 registers each): every wave draws
// work-list positions until the list is empty.  A position is a TICKET drawn when its wave is ready for it, not a
// block index: tickets are handed out in the order waves actually get to them, so everything a wave may wait for is
// held by a wave that is running (or done) whatever order the dispatcher picks — no co-residency assumption.  (One block
// per position, the first form, spent 0.14 ms per 10 000 positions on block launches alone.)
#ifndef IA3_FIT_LB
#define IA3_FIT_LB 2   // waves per SIMD the fit kernel is built for (256 registers at 2); scripts/ab_fit2.sh builds others
#endif
__global__ __launch_bounds__(64, IA3_FIT_LB) void fit_stages_k(FitArgs fa, int n, int stage0, int stage1, StageCtl* ctl,
                                                      int* done) {
  __shared__ WaveLds wl;
  IA3_LDS WaveLds* L = (IA3_LDS WaveLds*)&wl;
  const int lane = threadIdx.x & 63;
  // A fit is one wave working through dependent float64 chains: whenever it is ready it should issue.  Beside the
  // streaming kernels of other host threads (seven filter or warp waves on the same SIMD, round-robin) a fit that runs
  // to maxfev took three times as long as on an idle device, and every refit that waits for it waited with it.
  __builtin_amdgcn_s_setprio(3);
  // exact Voronoi ties in this field and no tie masks yet: nothing is fitted; the host builds the seed tree, resolves the
  // ties (voronoi_ties_k) and launches again
  // (the legacy model's driver, Fitting_v3.py:39-46, takes cdist + argmin instead of a tree: lowest index, no masks)
  if (stage0 == 0 && !fa.tie_lost && fa.variant == 0 && __builtin_amdgcn_readfirstlane(__hip_atomic_load(&ctl->ties, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
    if (lane == 0) __hip_atomic_fetch_max(&ctl->abort, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (lane == 0) { L->tally[0] = L->tally[1] = L->tally[2] = 0ull; L->tally_wait = 0ull; L->tally_conv = 0; L->tally_iter = 0; L->tally_fov = -1; }
  const unsigned long long t_start = __builtin_readcyclecounter();
#ifdef IA3_FIT_STAMPS
  if (lane < 24) { L->stamp[lane] = 0ull; L->stamp_fit[lane] = 0ull; }
  if (lane == 0) L->t_last = t_start;
#endif
  __builtin_amdgcn_wave_barrier();
  // one ticket: false when the counter is exhausted.  The load in front keeps exhausted counters free of atomics (the
  // value only grows, so "exhausted" is never a stale answer).
  auto draw = [&](StageCtl::Claim* c, unsigned size, unsigned& t) -> bool {
    if ((unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&c->next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= size) return false;
    unsigned v = 0;
    if (lane == 0) v = atomicAdd(&c->next, 1u);
    t = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
    IA3_STAMP(L, 0);   // ticket
    return t < size;
  };
  // ONE loop, one call site of the fit (the kernel is instruction-cache bound): first fits from the NCLAIM seed ranges,
  // own range first; then the refit sweeps in units of 64 positions, list order, each unit worked through in index order
  const int home = (int)(blockIdx.x % NCLAIM);
  const int later0 = stage0 > 1 ? stage0 : 1;
  // positions per unit of the refit sweeps: a power of two with n / (2 U) >= the positions expected to need a refit —
  // sweep 1 behind the first fits: the seeds with neighbours (counted per field by nbr_build_k); later sweeps: the
  // seeds not converged so far (all counts are final before this launch started: same value in every wave)
  int unit_size = 64;
  {
    long long needed;
    if (stage0 == 0) {   // seeds with neighbours, summed over the fields
      needed = 0;
      for (int f = 0; f < fa.n_fov; ++f) needed += (long long)fa.fov_counters[4 * (size_t)f + 3];
      needed = (long long)__builtin_amdgcn_readfirstlane((int)needed);
      if (!fa.fuse) needed = n;   // (test knob: every seed takes a separate sweep 1)
    } else {
      needed = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&ctl->n_unconv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    while (unit_size > 1 && needed * 2 * unit_size > (long long)n) unit_size >>= 1;
    // A launch from stage 0 is handed every sweep and keeps them all when a few seeds overlap others and most do not
    // (uint16 fields: the twin seeds of DoG plateaus, of which one refits noise for hundreds of evaluations in every
    // sweep): the work list orders the refits by their dependencies alone, so such a seed goes from sweep to sweep
    // without a launch boundary — a barrier over the whole batch — in between, and walking the positions of the
    // converged seeds, 64 at a time, costs microseconds.  Without overlaps (everything converges with sweep 1, the
    // walk would be pure cost) and in crowded fields (the walk goes position by position) it keeps the first two
    // stages, and the host sends the later sweeps in pairs while seeds are left.
    if (stage0 == 0 && stage1 > 2 && !(needed > 0 && needed * 64 <= (long long)n)) stage1 = 2;
    if (stage0 == 0 && blockIdx.x == 0 && lane == 0) ctl->stage_reached = stage1;
  }
  const unsigned per_stage = (unsigned)((n + unit_size - 1) / unit_size);
  const unsigned units = stage1 > later0 ? per_stage * (unsigned)(stage1 - later0) : 0u;
  int r = stage0 == 0 ? 0 : NCLAIM;   // next stage-0 range to try; NCLAIM: every first fit is in the hands of a running wave
  unsigned long long m = 0ull;        // positions of the current unit that still want a refit
  int uk = 0, ui0 = 0;
  for (;;) {
    int k, i;
    if (m) {
      const int b = __ffsll((long long)m) - 1;
      m &= m - 1;
      k = uk; i = ui0 + b;
    } else if (r < NCLAIM) {
      const int q = (home + r) % NCLAIM;
      const int lo = (int)((long long)n * q / NCLAIM), hi = (int)((long long)n * (q + 1) / NCLAIM);
      unsigned t;
      if (!draw(&ctl->claim[q], (unsigned)(hi - lo), t)) { ++r; continue; }
      k = 0; i = lo + (int)t;
    } else {
      unsigned u;
      if (!units || !draw(&ctl->claim[NCLAIM], units, u)) break;
      uk = later0 + (int)(u / per_stage);
      ui0 = (int)(u % per_stage) * unit_size;
      // every lane looks at one position of the unit: already made (a seed without neighbours gets sweep 1 with its
      // first fit) or skipped for good (converged, 1 << 20) -> nothing to do; the rest in index order
      bool need = false;
      if (lane < unit_size && ui0 + lane < n) need = __hip_atomic_load(&done[ui0 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < uk + 1;
      m = __ballot(need);
      continue;
    }
    if (!stage_position(fa, L, stage1, ctl, done, k, i)) break;
    __builtin_amdgcn_wave_barrier();
  }
  __builtin_amdgcn_wave_barrier();
  flush_tallies(fa, L);
  if (lane == 0) {   // (n_unconv: later positions of this launch only lose an early exit while the count is stale-high)
    if (L->tally_wait) atomicAdd(&fa.counters[3], L->tally_wait);
    atomicAdd(&fa.counters[4], __builtin_readcyclecounter() - t_start);   // wave cycles: the share of [3] is the wait share
    if (L->tally_conv) atomicSub(&ctl->n_unconv, L->tally_conv);
  }
#ifdef IA3_FIT_STAMPS
  if (lane < 24) atomicAdd(&fa.counters[8 + lane], L->stamp[lane]);   // the 256-byte counter slot holds 32 words
#endif
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

__global__ __launch_bounds__(64, 2) void fit_voxels_k(VoxArgs va, int n_fits, double ftol, double xtol, double gtol,
                                                      int maxfev, double factor) {
  __shared__ WaveLds wl;
  IA3_LDS WaveLds* L = (IA3_LDS WaveLds*)&wl;
  const int i = blockIdx.x;
  if (i >= n_fits) return;
  const int lane = threadIdx.x & 63;
  const int o0 = va.off[i], n = va.off[i + 1] - o0;
  unsigned valid = 0;
  const bool ok = n >= NP;
  {
    double vals[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int vi = lane + 64 * s;
      float fz = 0.f, fx = 0.f, fy = 0.f;
      vals[s] = 0.0;
      if (vi < n) {
        valid |= 1u << s;
        vals[s] = va.vals[o0 + vi];
        fz = (float)va.coords[3 * (o0 + vi)];
        fx = (float)va.coords[3 * (o0 + vi) + 1];
        fy = (float)va.coords[3 * (o0 + vi) + 2];
      }
      L->bl.dat[s][lane] = (float)vals[s]; L->bl.cz[s][lane] = fz; L->bl.cx[s][lane] = fx; L->bl.cy[s][lane] = fy;
    }
    if (ok) wave_extremes(vals, valid, L);
  }
  FitArgs fa;
  const double min_w = va.cfg[4 * i + 1], max_w = va.cfg[4 * i + 2];
  fa.min_ws = min_w * min_w; fa.max_ws = max_w * max_w; fa.init_w = va.cfg[4 * i + 3];
  fa.ftol = ftol; fa.xtol = xtol; fa.gtol = gtol; fa.maxfev = maxfev; fa.factor = factor;
  const double c0[3] = {va.center[3 * i], va.center[3 * i + 1], va.center[3 * i + 2]};
  if (lane < 11) L->p[lane] = NAN;
  __builtin_amdgcn_wave_barrier();
  int nfev = 0;
  if (ok) nfev = wave_gaussfit(fa, L, valid, va.kind[i], c0, va.cfg[4 * i], n, false);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 11; ++k) va.ps[(size_t)i * 11 + k] = L->p[k];
    for (int k = 0; k < NP; ++k) va.xs[(size_t)i * NP + k] = ok ? L->w.x[k] : NAN;
    va.info[2 * i] = ok ? 1 : 0;
    va.info[2 * i + 1] = nfev;
  }
}


// One launch prepares a fitter's pooled block: the zero-initialised arrays, NaN rows (all-ones float32: failed fits stay
// NaN rows, Fitting_v4.py:636), the stage control record, and — when the seeds never left the device — the seed
// coordinates.  (Nine separate memset / copy operations of ~5 us each before.)
struct InitArgs {
  uint4* zero0; size_t zero_words;      // 16-byte words to clear, head block excluded
  uint4* rows; size_t row_words;        // 16-byte words to fill with ones
  unsigned long long* counters;         // head block: [counters | n_iter | ctl | overflow], 256 bytes each
  int* niter; StageCtl* ctl; int* ovf;
  double* seeds; const double* src;     // optional device-to-device copy of 3 n doubles
  int n;                                // seeds — or, with n_dev, the capacity the block was laid out for
  // the seed count is still on the device (the seed stage's finish kernel leaves it there; the host learns it while this
  // kernel and the neighbour lists already run): seeds = min(*n_dev, n_cut if > 0), and nothing if that exceeds n
  const unsigned* n_dev; int n_cut;
  // one field of view: its three small tables are written here instead of uploaded (a copy from pageable memory in
  // front of the first fit launch costs the host ~20 us while the device idles)
  const void** ims; int* fov_start; int* fov_of; const void* im0;
};
__global__ __launch_bounds__(256) void fit_init_k(InitArgs a) {
  int n = a.n;
  if (a.n_dev) {
    long long m = (long long)*a.n_dev;
    if (a.n_cut > 0 && m > a.n_cut) m = a.n_cut;
    n = m <= (long long)a.n ? (int)m : 0;   // more seeds than the block holds: the host makes a fitter of the right size
  }
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
  for (size_t i = t; i < a.zero_words; i += step) a.zero0[i] = uint4{0u, 0u, 0u, 0u};
  for (size_t i = t; i < a.row_words; i += step) a.rows[i] = uint4{~0u, ~0u, ~0u, ~0u};
  if (a.src) for (size_t i = t; i < (size_t)3 * n; i += step) a.seeds[i] = a.src[i];
  if (t < 32) a.counters[t] = 0ull;
  if (t >= 128 && t < 192) a.niter[t - 128] = 0;   // one sweep counter per field (<= 64 fields per fitter)
  for (size_t k = t; k < sizeof(StageCtl) / 4; k += step) ((int*)a.ctl)[k] = k == 0 ? n : 0;   // n_unconv = n, all else 0
  if (t == 35) *a.ovf = 0;
  if (a.ims) {
    if (t == 36) a.ims[0] = a.im0;
    if (t == 37) a.fov_start[0] = 0;
    if (t == 38) a.fov_start[1] = n;
    for (size_t i = t; i < (size_t)a.n; i += step) a.fov_of[i] = 0;   // (the whole capacity: nbr_build_k's grid covers it)
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
using namespace ia3rt;

struct ia3_fitter {
  const ia3_stack* im;              // the first field's stack (shape and dtype of all of them)
  std::vector<const ia3_stack*> ims; // one per field of view
  std::vector<int> fov_start;        // n_fov + 1: seed ranges
  ia3_fit_params prm;
  int n;                             // seeds of all fields
  int nball;
  void *d_ims, *d_fov_start, *d_fov_of, *d_fov_ties, *d_fov_counters;
  std::vector<char> meta_stage;      // source of the asynchronous upload of the three tables above
  void* pool;          // one device block from the scratch cache holding every array below
  size_t pool_bytes;
  void *d_seeds, *d_nbr_cnt, *d_nbr_idx, *d_ball, *d_state, *d_ps, *d_nvox, *d_nfev, *d_memo, *d_conv, *d_niter,
      *d_counters, *d_done, *d_ctl, *d_nbr_overflow;
  bool pristine;       // the block is as fit_init_k left it: the first fit launch needs no further resets
  bool first_done;
  StageCtl host_ctl;
  unsigned long long host_counters[5];   // copy of d_counters as of the last ia3_fit_results(_ex): fits, evaluations, voxel
                                          // evaluations, shader cycles in dependency waits, wave cycles
  void *d_tie_flag, *d_tie_lost;   // per seed: has exact Voronoi ties (nbr_build_k) / lane masks of the tie voxels it loses
  void* kd_block;      // device copy of the seed trees (nodes | permutation per field), made only when ties exist
  bool ties_resolved;  // d_tie_lost is valid
  bool ties_host = false;   // ... and was made on the host (a device query overflowed its queue)
  std::vector<double> host_seeds;   // n x 3 when the seeds came from (or were fetched to) the host
  std::vector<char> kd_stage;       // source of the asynchronous tree upload
  bool cached;         // host_stage holds [counters | n_iter | ctl | overflow | rows] of the finished fit (run_sweeps)
  std::vector<char> host_stage;  // source of the asynchronous setup upload; lives as long as the fitter
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

int g_nb_cap = MAXNB;   // IA3_TUNE_FIT_NBLIST
int g_fit_fuse = 1;     // IA3_TUNE_FIT_FUSE: 1 = a seed without neighbours gets its first fit and sweep 1 from one wave
int g_fit_maxfev = 0;   // IA3_DEBUG_FIT_MAXFEV: profiling only (splits the kernel time into a fixed and a per-evaluation part)
int g_fit_waves = 2;    // IA3_TUNE_FIT_WAVES: persistent waves per SIMD (the kernel's 256 registers allow two)
int g_fit_merge = 1;    // IA3_TUNE_FIT_MERGE: sweeps after the first pair in one launch when few seeds are left (run_sweeps)
int g_fit_memo = 1;    // IA3_TUNE_FIT_MEMO: a refit whose neighbours have not changed since the seed's previous refit is not run
int g_fit_kdq = KDQ_CAP;   // IA3_TUNE_FIT_KDQ: queue entries per voxel of the device tie queries

FitArgs make_args(const ia3_fitter* f) {
  FitArgs a;
  a.ims = (const void* const*)f->d_ims; a.fov_of = (const int*)f->d_fov_of; a.fov_start = (const int*)f->d_fov_start;
  a.n_fov = (int)f->ims.size(); a.fov_counters = (unsigned long long*)f->d_fov_counters;
  a.dtype = f->im->dtype; a.Z = f->im->Z; a.X = f->im->X; a.Y = f->im->Y;
  a.n = f->n; a.fuse = g_fit_fuse; a.nb_cap = g_nb_cap; a.nb_r2 = 4.0 * f->prm.radius_fit * (double)f->prm.radius_fit;
  a.seeds = (const double*)f->d_seeds; a.nbr_cnt = (const int*)f->d_nbr_cnt; a.nbr_idx = (const int*)f->d_nbr_idx;
  a.ball = (const int*)f->d_ball; a.nball = f->nball; a.radius = f->prm.radius_fit;
  a.tie_lost = f->ties_resolved ? (const unsigned long long*)f->d_tie_lost : nullptr;
  a.state = (SeedState*)f->d_state; a.ps = (float*)f->d_ps; a.nvox = (int*)f->d_nvox; a.nfev = (int*)f->d_nfev;
  a.memo = g_fit_memo ? (unsigned long long*)f->d_memo : nullptr;
  a.conv = (unsigned char*)f->d_conv; a.n_iter = (int*)f->d_niter; a.counters = (unsigned long long*)f->d_counters;
  a.min_ws = f->prm.min_w * f->prm.min_w; a.max_ws = f->prm.max_w * f->prm.max_w; a.init_w = f->prm.init_w;
  a.delta_first = f->prm.min_delta_center; a.delta_repeat = f->prm.max_delta_center;
  a.dist_th2 = f->prm.max_dist_th * f->prm.max_dist_th;
  a.n_max_iter = f->prm.n_max_iter;
  // scipy.optimize.leastsq defaults used at Fitting_v4.py:388
  a.ftol = 1.49012e-8; a.xtol = 1.49012e-8; a.gtol = 0.0; a.maxfev = 1000; a.factor = 100.0;
  if (f->prm.model_variant == 1) {
    // Fitting_v3.py:253 calls leastsq without maxfev: MINPACK's default with Dfun is 100*(n+1)
    a.maxfev = 100 * (NP + 1);
    a.variant = 1;
    for (int k = 0; k < 3; ++k) {   // Fitting_v3.py:71-79 (the range test uses the un-squared bounds, as there)
      double w = f->prm.init_w_zxy[k];
      if (w * w > f->prm.max_w || w * w < f->prm.min_w) w = 1.5 * 1.5;
      a.iw[k] = log((a.max_ws - w * w) / (w * w - a.min_ws));
    }
  }
  if (g_fit_maxfev > 0) a.maxfev = g_fit_maxfev;
  return a;
}

}  // namespace

extern "C" {

// The staging buffer of a fitter (results on their way to the caller: 220 KB for 5 000 rows) is handed from one fitter of
// a thread to the next: a fresh std::vector of that size is an mmap, its page faults and an munmap per FOV — ~25 us on the
// host while the device has nothing to do.
static thread_local std::vector<char> t_spare_stage;
void ia3_fit_destroy(ia3_fitter* f) {
  if (!f) return;
  if (f->pool) ws_put(f->pool);   // back to the scratch cache; reuse is stream-ordered
  if (f->kd_block) ws_put(f->kd_block);
  if (f->host_stage.capacity() > t_spare_stage.capacity() && f->host_stage.capacity() <= (64u << 20)) t_spare_stage.swap(f->host_stage);
  delete f;
}

// The ball offset table depends on the radius only: uploaded once per process and radius, kept for good.
static int ball_table(int radius, const signed char** d_ball, int* nball) {
  struct Entry { pid_t pid; int radius, nball; void* d; };
  static std::vector<Entry> cache;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  for (auto& e : cache)
    if (e.pid == getpid() && e.radius == radius) { *d_ball = (const signed char*)e.d; *nball = e.nball; return IA3_OK; }
  std::vector<signed char> ball;
  const int nb = build_ball(radius, ball);
  if (nb > MAXBALL) return set_error(IA3_EUNSUPPORTED, "radius_fit %d gives %d voxels (> %d)", radius, nb, MAXBALL);
  void* d = nullptr;
  IA3_HIP(hipMalloc(&d, ball.size()));
  IA3_HIP(hipMemcpy(d, ball.data(), ball.size(), hipMemcpyHostToDevice));
  cache.push_back(Entry{getpid(), radius, nb, d});
  *d_ball = (const signed char*)d; *nball = nb;
  return IA3_OK;
}

constexpr int MAX_FOV = 64;   // fields of view per fitter (their sweep counters share one 256-byte slot)
struct FovSeeds { const ia3_stack* im; const double* host_zxy; const double* dev_zxy; int n; };

static int fit_create_impl(const FovSeeds* fovs, int n_fov, const ia3_fit_params* p, ia3_fitter** out,
                           const unsigned* n_dev = nullptr, int n_cut = 0) {
  int rc = ensure_init(); if (rc) return rc;
  dbg_stamp("fit_create enter");
  if (!fovs || n_fov < 1 || n_fov > MAX_FOV || !p || !out) return set_error(IA3_EINVAL, "bad argument");
  if (p->radius_fit < 1) return set_error(IA3_EINVAL, "radius_fit must be >= 1");
  long long ntot = 0;
  for (int k = 0; k < n_fov; ++k) {
    const FovSeeds& q = fovs[k];
    if (!q.im || q.n < 0 || (q.n > 0 && !q.host_zxy && !q.dev_zxy)) return set_error(IA3_EINVAL, "bad argument");
    if (q.im->dtype != fovs[0].im->dtype || q.im->Z != fovs[0].im->Z || q.im->X != fovs[0].im->X || q.im->Y != fovs[0].im->Y)
      return set_error(IA3_EINVAL, "the fields of one fitter must have the same shape and dtype");
    if (q.host_zxy)
      for (int i = 0; i < 3 * q.n; ++i)
        if (!(fabs(q.host_zxy[i]) < 1e9)) return set_error(IA3_EINVAL, "non-finite seed coordinate");
    ntot += q.n;
  }
  if (ntot > 0x3fffffff) return set_error(IA3_EUNSUPPORTED, "too many seeds");
  const int n = (int)ntot;
  const ia3_stack* im = fovs[0].im;
  const signed char* d_ball = nullptr;
  int nball = 0;
  rc = ball_table(p->radius_fit, &d_ball, &nball); if (rc) return rc;
  ia3_fitter* f = new ia3_fitter();   // value-initialised: pointers null, flags false
  f->host_stage.swap(t_spare_stage);   // capacity of the thread's previous fitter (contents are overwritten before use)
  f->host_stage.clear();
  f->im = im; f->prm = *p; f->n = n; f->nball = nball;
  f->fov_start.assign(1, 0);
  for (int k = 0; k < n_fov; ++k) { f->ims.push_back(fovs[k].im); f->fov_start.push_back(f->fov_start.back() + fovs[k].n); }
  // one pooled device block: [uploaded read-only part | zero-initialised part | NaN-initialised rows | neighbour lists]
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t b_seeds = al(sizeof(double) * 3 * (size_t)n);
  const size_t b_ims = al(sizeof(void*) * (size_t)n_fov), b_fstart = al(sizeof(int) * (size_t)(n_fov + 1)), b_fof = al(sizeof(int) * (size_t)n);
  const size_t up_bytes = b_seeds + b_ims + b_fstart + b_fof;
  const size_t b_state = al(sizeof(SeedState) * (size_t)n), b_nvox = al(sizeof(int) * (size_t)n), b_nfev = b_nvox,
               b_conv = al((size_t)n), b_niter = 256, b_cnt = 256, b_done = al(sizeof(int) * (size_t)n), b_ctl = al(sizeof(StageCtl)),
               b_ovf = 256, b_fties = al(sizeof(int) * (size_t)n_fov), b_fcnt = al(4 * sizeof(unsigned long long) * (size_t)n_fov);
  const size_t b_memo = al(sizeof(unsigned long long) * (size_t)n);
  const size_t zero_bytes = b_state + b_nvox + b_nfev + b_memo + b_conv + b_niter + b_cnt + b_done + b_ctl + b_ovf + b_fties + b_fcnt;
  const size_t b_ps = al(sizeof(float) * 11 * (size_t)n);
  const size_t b_ncnt = al(sizeof(int) * (size_t)n), b_nidx = al(sizeof(int) * MAXNB * (size_t)n);
  const size_t b_tflag = al(sizeof(int) * (size_t)n), b_tlost = al(sizeof(unsigned long long) * SLOTS * (size_t)n);
  f->pool_bytes = up_bytes + zero_bytes + b_ps + b_ncnt + b_nidx + b_tflag + b_tlost;
  f->pool = ws_get(f->pool_bytes);
  if (!f->pool) { delete f; return IA3_ENOMEM; }
  char* base = (char*)f->pool;
  size_t o = 0;
  f->d_seeds = base + o; o += b_seeds;
  char* meta0 = base + o;
  f->d_ims = base + o; o += b_ims;
  f->d_fov_start = base + o; o += b_fstart;
  f->d_fov_of = base + o; o += b_fof;
  f->d_ball = (void*)d_ball;
  char* zero0 = base + o;
  f->d_state = base + o; o += b_state;
  f->d_nvox = base + o; o += b_nvox;
  f->d_nfev = base + o; o += b_nfev;
  f->d_memo = base + o; o += b_memo;
  f->d_conv = base + o; o += b_conv;
  f->d_done = base + o; o += b_done;
  f->d_fov_ties = base + o; o += b_fties;
  f->d_fov_counters = base + o; o += b_fcnt;
  // [counters | n_iter | stage control | overflow flag | rows]: contiguous, so the results come back in ONE
  // device-to-host copy (every separate copy into pageable memory costs a ~25 us round trip)
  f->d_counters = base + o; o += b_cnt;
  f->d_niter = base + o; o += b_niter;
  f->d_ctl = base + o; o += b_ctl;
  f->d_nbr_overflow = base + o; o += b_ovf;
  f->d_ps = base + o; o += b_ps;
  f->d_nbr_cnt = base + o; o += b_ncnt;
  f->d_nbr_idx = base + o; o += b_nidx;
  f->d_tie_flag = base + o; o += b_tflag;
  f->d_tie_lost = base + o;
  hipStream_t st = stream();
  hipError_t e = hipSuccess;
  // the three small tables: stack pointers, seed ranges, field of every seed
  if (n_fov > 1) {
    std::vector<char>& m = f->meta_stage;
    m.assign(b_ims + b_fstart + b_fof, 0);
    for (int k = 0; k < n_fov; ++k) { const void* d = fovs[k].im->d; memcpy(m.data() + sizeof(void*) * (size_t)k, &d, sizeof(void*)); }
    memcpy(m.data() + b_ims, f->fov_start.data(), sizeof(int) * (size_t)(n_fov + 1));
    int* fof = (int*)(m.data() + b_ims + b_fstart);
    for (int k = 0; k < n_fov; ++k) for (int i = f->fov_start[k]; i < f->fov_start[k + 1]; ++i) fof[i] = k;
    e = hipMemcpyAsync(meta0, m.data(), m.size(), hipMemcpyHostToDevice, st);
  }
  // seeds: from the host (staged in the fitter: the copy is asynchronous) or already resident (device-to-device)
  bool all_host = true, any_host = false;
  for (int k = 0; k < n_fov; ++k) { if (fovs[k].n) { if (fovs[k].host_zxy) any_host = true; else all_host = false; } }
  if (any_host) {
    std::vector<char>& host = f->host_stage;
    host.assign(sizeof(double) * 3 * (size_t)n, 0);
    for (int k = 0; k < n_fov; ++k)
      if (fovs[k].n && fovs[k].host_zxy)
        memcpy(host.data() + sizeof(double) * 3 * (size_t)f->fov_start[k], fovs[k].host_zxy, sizeof(double) * 3 * (size_t)fovs[k].n);
    if (all_host) f->host_seeds.assign((const double*)host.data(), (const double*)host.data() + 3 * (size_t)n);
  }
  for (int k = 0; k < n_fov && e == hipSuccess; ++k) {
    if (!fovs[k].n) continue;
    char* dst = (char*)f->d_seeds + sizeof(double) * 3 * (size_t)f->fov_start[k];
    const size_t bytes = sizeof(double) * 3 * (size_t)fovs[k].n;
    if (fovs[k].host_zxy)
      e = hipMemcpyAsync(dst, f->host_stage.data() + sizeof(double) * 3 * (size_t)f->fov_start[k], bytes, hipMemcpyHostToDevice, st);
    else if (n_fov > 1)   // (a single resident list is copied by fit_init_k itself: one launch less on the per-FOV path)
      e = hipMemcpyAsync(dst, fovs[k].dev_zxy, bytes, hipMemcpyDeviceToDevice, st);
  }
  if (e != hipSuccess) { ia3_fit_destroy(f); return set_error(IA3_EHIP, "fitter setup failed: %s", hipGetErrorString(e)); }
  {
    InitArgs ia;
    const size_t head0 = (size_t)((char*)f->d_counters - zero0);
    ia.zero0 = (uint4*)zero0; ia.zero_words = head0 / 16;           // [state .. field counters]; the head block is set by name
    ia.rows = (uint4*)f->d_ps; ia.row_words = b_ps / 16;
    ia.counters = (unsigned long long*)f->d_counters; ia.niter = (int*)f->d_niter; ia.ctl = (StageCtl*)f->d_ctl;
    ia.ovf = (int*)f->d_nbr_overflow;
    ia.seeds = (double*)f->d_seeds; ia.src = (n_fov == 1 && !fovs[0].host_zxy) ? fovs[0].dev_zxy : nullptr;
    ia.n = n;
    ia.n_dev = n_dev; ia.n_cut = n_cut;
    ia.ims = n_fov == 1 ? (const void**)f->d_ims : nullptr;
    ia.fov_start = (int*)f->d_fov_start; ia.fov_of = (int*)f->d_fov_of; ia.im0 = fovs[0].im->d;
    size_t words = ia.zero_words > ia.row_words ? ia.zero_words : ia.row_words;
    unsigned blocks = (unsigned)((words + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    dbg_stamp("fit_init launch");
    hipLaunchKernelGGL(fit_init_k, dim3(blocks), dim3(256), 0, st, ia);
    dbg_stamp("fit_init launched");
  }
  f->pristine = true;
  if (n > 0) {
    const double rr = 2.0 * p->radius_fit;
    ProfScope ps("nbr_build");
    hipLaunchKernelGGL(nbr_build_k, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, (const double*)f->d_seeds, n, rr * rr,
                       (int*)f->d_nbr_cnt, (int*)f->d_nbr_idx, (int*)f->d_nbr_overflow, (const int*)d_ball, nball,
                       im->Z, im->X, im->Y, (int*)f->d_tie_flag, &((StageCtl*)f->d_ctl)->ties,
                       (const int*)f->d_fov_of, (const int*)f->d_fov_start, (int*)f->d_fov_ties,
                       (unsigned long long*)f->d_fov_counters);
  }
  {
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) { ia3_fit_destroy(f); return set_error(IA3_EHIP, "fitter setup launch failed: %s", hipGetErrorString(le)); }
  }
  *out = f;
  return IA3_OK;
}

int ia3_fit_create(const ia3_stack* im, const double* centers_zxy, int n, const ia3_fit_params* p,
                   ia3_fitter** out) {
  if (!im || n < 0 || (n > 0 && !centers_zxy)) return set_error(IA3_EINVAL, "bad argument");
  const FovSeeds one{im, centers_zxy, nullptr, n};
  return fit_create_impl(&one, 1, p, out);
}

}  // extern "C"

namespace ia3k {
void set_fit_nblist(int cap) { g_nb_cap = cap < 0 ? 0 : (cap > MAXNB ? MAXNB : cap); }
void set_fit_fuse(int on) { g_fit_fuse = on ? 1 : 0; }
void set_fit_maxfev(int n) { g_fit_maxfev = n; }
void set_fit_waves(int n) { g_fit_waves = n < 1 ? 1 : (n > IA3_FIT_LB ? IA3_FIT_LB : n); }
void set_fit_merge(int on) { g_fit_merge = on != 0; }
int set_fit_waitbound(int polls) {
  int rc = ia3rt::ensure_init(); if (rc) return rc;
  const long long v = polls > 0 ? (long long)polls : (1LL << 22);
  IA3_HIP(hipMemcpyToSymbol(HIP_SYMBOL(d_wait_bound), &v, sizeof(v)));
  return IA3_OK;
}
void set_fit_memo(int on) { g_fit_memo = on != 0; }
void set_fit_kdq(int cap) { g_fit_kdq = cap < 1 ? 1 : (cap > KDQ_CAP ? KDQ_CAP : cap); }
void fit_host_counters(const ia3_fitter* f, long long out[5]) {
  for (int k = 0; k < 5; ++k) out[k] = (long long)f->host_counters[k];
}
int fit_create_dev(const ia3_stack* im, const double* d_centers_zxy, int n, const ia3_fit_params* p, ia3_fitter** out) {
  if (!im || n < 0 || (n > 0 && !d_centers_zxy)) return set_error(IA3_EINVAL, "bad argument");
  const FovSeeds one{im, nullptr, d_centers_zxy, n};
  return fit_create_impl(&one, 1, p, out);
}
// A fitter made BEFORE the host knows the seed count: laid out for `capacity` seeds, its set-up kernels (fit_init_k,
// nbr_build_k) read the count the seed stage leaves on the device (min(*d_count, n_cut if > 0)) and are queued right
// behind it; fit_set_count gives the host side the number once it has arrived.  (The host used to learn the count first
// and queue the three launches afterwards: 45-70 us of idle device between the seed finish and the fit, kernel trace of
// profiles/r04j.)  A count above the capacity: destroy this fitter and make one the ordinary way.
int fit_create_ahead(const ia3_stack* im, const double* d_centers_zxy, int capacity, const unsigned* d_count, int n_cut,
                     const ia3_fit_params* p, ia3_fitter** out) {
  if (!im || capacity < 1 || !d_centers_zxy || !d_count) return set_error(IA3_EINVAL, "bad argument");
  const FovSeeds one{im, nullptr, d_centers_zxy, capacity};
  return fit_create_impl(&one, 1, p, out, d_count, n_cut);
}
int fit_set_count(ia3_fitter* f, int n) {
  if (!f || n < 0 || n > f->n || f->ims.size() != 1) return set_error(IA3_EINVAL, "bad seed count");
  f->n = n;
  f->fov_start.assign({0, n});
  return IA3_OK;
}
// one fitter over several fields of view (same shape and dtype, at most fit_max_fovs() of them): seeds resident per field
int fit_max_fovs() { return MAX_FOV; }
int fit_create_multi(const ia3_stack* const* ims, const double* const* d_centers_zxy, const int* n_seeds, int n_fov,
                     const ia3_fit_params* p, ia3_fitter** out) {
  std::vector<FovSeeds> v((size_t)(n_fov > 0 ? n_fov : 0));
  for (int k = 0; k < n_fov; ++k) v[(size_t)k] = FovSeeds{ims[k], nullptr, d_centers_zxy[k], n_seeds[k]};
  return fit_create_impl(v.data(), n_fov, p, out);
}
// after ia3_fit_results(_ex): per field its sweep count and its fits / evaluations / voxel evaluations
int fit_fov_results(ia3_fitter* f, int* n_iter, long long* counters3) {
  const int nf = (int)f->ims.size();
  std::vector<int> it((size_t)nf);
  std::vector<unsigned long long> c(4 * (size_t)nf);
  IA3_HIP(hipStreamSynchronize(stream()));
  IA3_HIP(hipMemcpy(it.data(), f->d_niter, sizeof(int) * (size_t)nf, hipMemcpyDeviceToHost));
  IA3_HIP(hipMemcpy(c.data(), f->d_fov_counters, sizeof(unsigned long long) * 4 * (size_t)nf, hipMemcpyDeviceToHost));
  for (int k = 0; k < nf; ++k) {
    if (n_iter) n_iter[k] = it[(size_t)k];
    if (counters3) for (int j = 0; j < 3; ++j) counters3[3 * k + j] = (long long)c[4 * (size_t)k + j];
  }
  return IA3_OK;
}
const int* fit_fov_starts(const ia3_fitter* f) { return f->fov_start.data(); }
}  // namespace ia3k

extern "C" {

// Launch the work list for stages [stage0, stage1).  `fresh` re-arms the whole control block (claim = 0,
// n_unconv = n, abort = 0); otherwise only the ticket counter is reset and n_unconv / done[] carry over.
static int launch_stages(ia3_fitter* f, int stage0, int stage1, bool fresh) {
  FitArgs a = make_args(f);
  hipStream_t st = stream();
  if (fresh) {
    if (!f->pristine) {   // fit_init_k has armed the control record of a new fitter already
      // n_unconv = n, abort = 0; the finding of nbr_build_k (ties) stays
      memset(&f->host_ctl, 0, sizeof(StageCtl));
      f->host_ctl.n_unconv = f->n;
      static_assert(offsetof(StageCtl, n_unconv) == 0 && offsetof(StageCtl, abort) == 4, "the first two words are re-armed together");
      IA3_HIP(hipMemcpyAsync(f->d_ctl, &f->host_ctl, 2 * sizeof(int), hipMemcpyHostToDevice, st));
    }
  }
  if (!fresh || !f->pristine)
    IA3_HIP(hipMemsetAsync((char*)f->d_ctl + offsetof(StageCtl, claim), 0, sizeof(StageCtl) - offsetof(StageCtl, claim), st));   // the ticket counters
  f->pristine = false;
  long long blocks = (long long)(stage1 - stage0) * f->n;   // work-list positions; waves draw them as tickets
  const long long simds = 4LL * num_cus() * g_fit_waves;
  if (blocks > simds) blocks = simds;
  if (blocks < 1) blocks = 1;
  ProfScope ps(stage0 == 0 ? "fit_first" : "fit_repeat");
  hipLaunchKernelGGL(fit_stages_k, dim3((unsigned)blocks), dim3(64), 0, st, a, f->n, stage0, stage1, (StageCtl*)f->d_ctl,
                     (int*)f->d_done);
  dbg_stamp("fit_stages launched");
  IA3_KCHECK();
  return IA3_OK;
}

// [counters | n_iter | stage control | overflow flag | rows] -> f->host_stage, one wait.
// The block is written into the thread's pinned, device-mapped mailbox by a KERNEL queued behind the fit, followed by
// a sequence word the host polls.  A hipMemcpyAsync queued behind the fit parks a barrier on a copy engine's ring
// until the fit kernel has ended, and every other stream's device-to-host copy that lands on that ring waits with it:
// a field with plateau twins (a 30 ms launch) held up the seed read-backs of the images seeded beside it, so three
// images "in flight" were fitted one after the other (rocprofv3 API + kernel trace of the movie leg, profiles/r03d).

__global__ __launch_bounds__(1024) void mail_copy_k(const unsigned* __restrict__ src, unsigned* __restrict__ dst, size_t n4,
                                                    volatile unsigned* seq_word, unsigned seq) {
  const size_t n16 = n4 / 4;
  for (size_t i = threadIdx.x; i < n16; i += 1024) ((uint4*)dst)[i] = ((const uint4*)src)[i];
  for (size_t i = 4 * n16 + threadIdx.x; i < n4; i += 1024) dst[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) *seq_word = seq;
}
static int fetch_block(ia3_fitter* f, bool with_rows) {
  hipStream_t st = stream();
  const size_t head = (size_t)((char*)f->d_ps - (char*)f->d_counters);
  const size_t rows = with_rows ? sizeof(float) * 11 * (size_t)f->n : 0;
  f->host_stage.resize(head + rows);
  void *mh = nullptr, *md = nullptr;
  constexpr size_t MAIL_OFF = 4096;   // the first page holds the control words: the seed stage's at 0, the fit's sequence word at 2048
  constexpr size_t SEQ_OFF = 2048;
  static_assert(sizeof(uint4) == 16, "");
  if (head + rows <= (1u << 20) - MAIL_OFF && (head + rows) % 4 == 0 && ((size_t)f->d_counters & 15) == 0 &&
      host_mailbox(1u << 20, &mh, &md) == IA3_OK) {
    static thread_local unsigned t_seq = 0;
    const unsigned seq = ++t_seq ? t_seq : ++t_seq;   // never 0 (the mailbox starts zeroed)
    hipLaunchKernelGGL(mail_copy_k, dim3(1), dim3(1024), 0, st, (const unsigned*)f->d_counters, (unsigned*)((char*)md + MAIL_OFF),
                       (head + rows) / 4, (volatile unsigned*)((char*)md + SEQ_OFF), seq);
    IA3_KCHECK();
    // scratch of the seed stage (PutDefer in ia3_fit_fov_dev) goes back to the cache now: its event records queue up
    // behind the copy the host is about to wait for, not in front of it
    ws_put_deferred_now();
    volatile unsigned* mb = (volatile unsigned*)((char*)mh + SEQ_OFF);
    SpinWait sw;
    while (*mb != seq) {
      sw.relax();
      if (((sw.n & 0xfffff) == 0 || (sw.n > 40400 && (sw.n & 0x3ff) == 0)) && hipStreamQuery(st) != hipErrorNotReady) {   // the stream drained (or failed) without the word
        if (*mb == seq) break;
        IA3_HIP(hipStreamSynchronize(st));
        if (*mb != seq) return set_error(IA3_EHIP, "fit results did not reach the host mailbox");
        break;
      }
    }
    memcpy(f->host_stage.data(), (char*)mh + MAIL_OFF, head + rows);
    return IA3_OK;
  }
  // too large for the mailbox: wait for the fit first, copy afterwards (nothing parked on a copy engine meanwhile)
  { int rc = stream_wait_spin(st); if (rc) return rc; }
  IA3_HIP(hipMemcpyAsync(f->host_stage.data(), f->d_counters, head + rows, hipMemcpyDeviceToHost, st));
  IA3_HIP(hipStreamSynchronize(st));
  return IA3_OK;
}

// Exact Voronoi ties (nbr_build_k flagged them, the first launch left without fitting anything): build the seed tree with
// cKDTree's layout on the host (kdtree.cpp), hand it to voronoi_ties_k, which leaves per seed the lane masks of the tie
// voxels it loses.  Fields without ties — every isolated-spot field — never come here.
static int resolve_ties_host(ia3_fitter* f);
static int resolve_ties(ia3_fitter* f) {
  hipStream_t st = stream();
  const int n = f->n, nf = (int)f->ims.size();
  std::vector<int> flagged((size_t)nf, 1);
  if (f->host_seeds.empty()) {   // the seed list never left the device: fetch it now, with the per-field tie flags
    f->host_seeds.resize(3 * (size_t)n);
    IA3_HIP(hipMemcpyAsync(f->host_seeds.data(), f->d_seeds, sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToHost, st));
  }
  IA3_HIP(hipMemcpyAsync(flagged.data(), f->d_fov_ties, sizeof(int) * (size_t)nf, hipMemcpyDeviceToHost, st));
  IA3_HIP(hipStreamSynchronize(st));
  // one tree per field that has ties (cKDTree(self.centers) of that image, Fitting_v4.py:601)
  struct Built { int fov; std::vector<ia3::KdNode> nodes; std::vector<int> perm; KdTree t; size_t off_nodes, off_perm; };
  std::vector<Built> trees;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t total = 0;
  for (int k = 0; k < nf; ++k) {
    const int s0 = f->fov_start[(size_t)k], nk = f->fov_start[(size_t)k + 1] - s0;
    if (!flagged[(size_t)k] || nk == 0) continue;
    trees.emplace_back();
    Built& b = trees.back();
    b.fov = k;
    ia3k::kd_build(f->host_seeds.data() + 3 * (size_t)s0, nk, b.nodes, b.perm, b.t.mins, b.t.maxes);
    b.off_nodes = total; total += al(b.nodes.size() * sizeof(ia3::KdNode));
    b.off_perm = total; total += al(b.perm.size() * sizeof(int));
  }
  if (f->kd_block) { ws_put(f->kd_block); f->kd_block = nullptr; }
  f->kd_block = ws_get(total ? total : 256);
  if (!f->kd_block) return IA3_ENOMEM;
  f->kd_stage.assign(total, 0);
  for (const Built& b : trees) {
    memcpy(f->kd_stage.data() + b.off_nodes, b.nodes.data(), b.nodes.size() * sizeof(ia3::KdNode));
    memcpy(f->kd_stage.data() + b.off_perm, b.perm.data(), b.perm.size() * sizeof(int));
  }
  if (total) IA3_HIP(hipMemcpyAsync(f->kd_block, f->kd_stage.data(), total, hipMemcpyHostToDevice, st));
  FitArgs a = make_args(f);
  {
    ProfScope ps("voronoi_ties");
    for (Built& b : trees) {
      const int s0 = f->fov_start[(size_t)b.fov], nk = f->fov_start[(size_t)b.fov + 1] - s0;
      b.t.nodes = (const ia3::KdNode*)((char*)f->kd_block + b.off_nodes);
      b.t.indices = (const int*)((char*)f->kd_block + b.off_perm);
      b.t.data = (const double*)f->d_seeds + 3 * (size_t)s0;
      b.t.n = nk;
      hipLaunchKernelGGL(voronoi_ties_k, dim3((unsigned)nk), dim3(64), 0, st, a, b.t, s0, (const int*)f->d_tie_flag,
                         (unsigned long long*)f->d_tie_lost, &((StageCtl*)f->d_ctl)->abort, g_fit_kdq);
    }
  }
  IA3_KCHECK();
  // did a query run out of queue entries?  (The next fit launch re-arms the control record, so this is the place to
  // look; fields with ties are the rare case and pay one more small read-back.)
  int ab = 0;
  IA3_HIP(hipMemcpyAsync(&ab, &((StageCtl*)f->d_ctl)->abort, sizeof(int), hipMemcpyDeviceToHost, st));
  IA3_HIP(hipStreamSynchronize(st));
  f->ties_resolved = true;
  if (ab == 3) return resolve_ties_host(f);
  return IA3_OK;
}

// The same decisions on the host, for the case that a device query overflowed its queue (KDQ_CAP entries per voxel; never
// seen on real fields): per field with ties the tree is still in f->kd_stage, every voxel of a flagged seed's ball asks
// it through a queue as large as the tree, and a voxel is lost in a tie iff the tree answers with another seed that is
// exactly as far away (voronoi_ties_k's rule: a strictly nearer seed is not a tie and is handled by the gather itself).
static int resolve_ties_host(ia3_fitter* f) {
  hipStream_t st = stream();
  const int n = f->n, nf = (int)f->ims.size();
  std::vector<int> flag((size_t)n), fov_ties((size_t)nf);
  std::vector<int> ball((size_t)f->nball);
  IA3_HIP(hipMemcpyAsync(flag.data(), f->d_tie_flag, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, st));
  IA3_HIP(hipMemcpyAsync(fov_ties.data(), f->d_fov_ties, sizeof(int) * (size_t)nf, hipMemcpyDeviceToHost, st));
  IA3_HIP(hipMemcpyAsync(ball.data(), f->d_ball, sizeof(int) * (size_t)f->nball, hipMemcpyDeviceToHost, st));
  IA3_HIP(hipStreamSynchronize(st));
  std::vector<unsigned long long> lost((size_t)n * SLOTS, 0ull);
  auto d2 = [](const double* a, double z, double x, double y) {   // dist2_seq: separate multiplies and adds, in this order
    const double d0 = a[0] - z, d1 = a[1] - x, d2_ = a[2] - y;
    volatile double m0 = d0 * d0, m1 = d1 * d1, m2 = d2_ * d2_;
    volatile double s01 = m0 + m1;
    return (double)(s01 + m2);
  };
  const double r2x = 2.0 * (double)f->prm.radius_fit;
  for (int k = 0; k < nf; ++k) {
    const int s0 = f->fov_start[(size_t)k], nk = f->fov_start[(size_t)k + 1] - s0;
    if (!fov_ties[(size_t)k] || nk == 0) continue;
    std::vector<ia3::KdNode> nodes;
    std::vector<int> perm;
    KdTree t;
    ia3k::kd_build(f->host_seeds.data() + 3 * (size_t)s0, nk, nodes, perm, t.mins, t.maxes);
    t.nodes = nodes.data(); t.indices = perm.data(); t.data = f->host_seeds.data() + 3 * (size_t)s0; t.n = nk;
    std::vector<KdQEntry> heap(nodes.size() + 2);
    const ia3_stack* im = f->ims[(size_t)k];
    for (int i = s0; i < s0 + nk; ++i) {
      if (!flag[(size_t)i]) continue;
      const double* c0 = f->host_seeds.data() + 3 * (size_t)i;
      const int iz = (int)c0[0], ix = (int)c0[1], iy = (int)c0[2];
      for (int vi = 0; vi < f->nball; ++vi) {
        const int w_ = ball[(size_t)vi];
        const int z = iz + (int)(signed char)(w_ & 0xff), x = ix + (int)(signed char)((w_ >> 8) & 0xff), y = iy + (int)(signed char)((w_ >> 16) & 0xff);
        if (z < 0 || z >= im->Z || x < 0 || x >= im->X || y < 0 || y >= im->Y) continue;
        const double q[3] = {(double)z, (double)x, (double)y};
        KdQueue<KdQEntry*> queue(heap.data(), 1, (int)heap.size());
        const int w = kd_nearest(t, q, r2x, queue);
        if (queue.overflow) return set_error(IA3_EUNSUPPORTED, "Voronoi tie query exceeded the host queue");
        if (w < 0 || w == i - s0) continue;
        const double* cw = f->host_seeds.data() + 3 * (size_t)(s0 + w);
        if (d2(cw, q[0], q[1], q[2]) == d2(c0, q[0], q[1], q[2])) lost[(size_t)i * SLOTS + (size_t)(vi / 64)] |= 1ull << (vi % 64);
      }
    }
  }
  f->kd_stage.assign((const char*)lost.data(), (const char*)lost.data() + lost.size() * sizeof(unsigned long long));
  IA3_HIP(hipMemcpyAsync(f->d_tie_lost, f->kd_stage.data(), f->kd_stage.size(), hipMemcpyHostToDevice, st));
  IA3_HIP(hipStreamSynchronize(st));   // kd_stage is reused; the upload is small
  f->ties_host = true;
  return IA3_OK;
}

static int check_ctl(const StageCtl& hc) {
  if (hc.abort == 3) return set_error(IA3_EUNSUPPORTED, "Voronoi tie query exceeded its queue (%d entries per voxel)", KDQ_CAP);
  if (hc.abort) return set_error(IA3_EHIP, "fit kernel aborted: a dependency wait exceeded its bound");
  return IA3_OK;
}

// Sweeps are launched two at a time and the next pair only while some seed is still unconverged: a launch costs
// one block per (stage, seed) even for converged seeds, and most fields converge after the first sweep.  The check
// between pairs fetches the row table along with the control words: when nothing is left to refit (the common case)
// ia3_fit_results finds everything on the host already and the fit costs ONE synchronisation.
static int run_sweeps(ia3_fitter* f, int stage, bool fresh, int last = -1) {
  if (last < 0) last = f->prm.n_max_iter + 2;   // sweeps 1 .. n_max_iter+1 (Fitting_v4.py:683)
  f->cached = false;
  bool few_left = false;   // after the first pair: a handful of seeds still move (plateau twins that refit noise, a few chains)
  while (stage < last) {
    // ... then every remaining sweep goes out in ONE launch: the work list orders the refits by their true dependencies,
    // so a seed's sweep k+1 follows its own sweep k instead of the slowest fit of sweep k anywhere in the batch (a launch
    // boundary is a barrier over all fields; three uint16 fields took 83 ms in pairs of sweeps).  The positions of
    // converged seeds are skipped 64 at a time (unit_size in fit_stages_k), which is only cheap while few seeds are left:
    // crowded fields keep the pairs.
    const bool kernel_decides = stage == 0 && g_fit_merge && last > 2;   // (fit_stages_k: all sweeps, or the first two stages)
    int s1 = few_left || kernel_decides ? last : (stage + 2 < last ? stage + 2 : last);
    int rc = launch_stages(f, stage, s1, fresh); if (rc) return rc;
    const bool with_first = stage == 0 && !f->ties_resolved;   // this launch may have left at once: exact Voronoi ties
    if (s1 >= last && !with_first && !kernel_decides) break;
    const bool rows_now = f->n <= 16384;   // a batch of fields: the (large) row table is fetched once, at the end
    rc = fetch_block(f, rows_now); if (rc) return rc;
    StageCtl hc;
    memcpy(&hc, f->host_stage.data() + ((char*)f->d_ctl - (char*)f->d_counters), sizeof(StageCtl));
    if (hc.abort == 2 && with_first) {
      rc = resolve_ties(f); if (rc) return rc;
      fresh = true;
      continue;   // the same stages again, now with the tie masks
    }
    rc = check_ctl(hc); if (rc) return rc;
    fresh = false;
    if (kernel_decides) s1 = hc.stage_reached;
    stage = s1;
    if (stage >= last) { f->cached = rows_now; break; }
    if (hc.n_unconv <= 0) { f->cached = rows_now; break; }
    few_left = g_fit_merge && (long long)hc.n_unconv * 64 <= (long long)f->n;
  }
  return IA3_OK;
}

int ia3_fit_first(ia3_fitter* f) {
  if (!f) return set_error(IA3_EINVAL, "null fitter");
  f->cached = false;
  if (f->n > 0) {
    if (!f->pristine) IA3_HIP(hipMemsetAsync(f->d_done, 0, sizeof(int) * (size_t)f->n, stream()));
    int rc = run_sweeps(f, 0, true, 1); if (rc) return rc;
    f->cached = false;   // ia3_fit_repeat follows
  }
  f->first_done = true;
  return IA3_OK;
}

int ia3_fit_repeat(ia3_fitter* f, int* n_iter) {
  if (!f) return set_error(IA3_EINVAL, "null fitter");
  if (!f->first_done) return set_error(IA3_EINVAL, "repeatfit() before firstfit()");
  int it = 0;
  if (f->n > 0) {
    hipStream_t st = stream();
    IA3_HIP(hipMemset2DAsync((char*)f->d_state + offsetof(SeedState, conv), sizeof(SeedState), 0, sizeof(int), (size_t)f->n, st));
    IA3_HIP(hipMemsetAsync(f->d_niter, 0, sizeof(int), st));
    IA3_HIP(hipMemsetAsync(f->d_memo, 0, sizeof(unsigned long long) * (size_t)f->n, st));   // a new repeatfit() refits every seed
    IA3_HIP(hipMemsetD32Async((hipDeviceptr_t)f->d_done, 1, (size_t)f->n, st));   // every seed has completed stage 0
    int rc = run_sweeps(f, 1, true); if (rc) return rc;
    if (n_iter) {   // callers that fetch results next pass NULL and read n_iter with them (one sync)
      IA3_HIP(hipMemcpyAsync(&it, f->d_niter, sizeof(int), hipMemcpyDeviceToHost, st));
      IA3_HIP(hipStreamSynchronize(st));
    }
  }
  if (n_iter) *n_iter = it;
  return IA3_OK;
}

// firstfit + repeatfit: stage 0 and sweep 1 go out in ONE launch (no host round trip between them)
int ia3_fit_run(ia3_fitter* f) {
  if (!f) return set_error(IA3_EINVAL, "null fitter");
  if (f->n > 0) {
    hipStream_t st = stream();
    if (!f->pristine) {
      IA3_HIP(hipMemsetAsync(f->d_done, 0, sizeof(int) * (size_t)f->n, st));
      IA3_HIP(hipMemset2DAsync((char*)f->d_state + offsetof(SeedState, conv), sizeof(SeedState), 0, sizeof(int), (size_t)f->n, st));
      IA3_HIP(hipMemsetAsync(f->d_niter, 0, sizeof(int), st));
    }
    int rc = run_sweeps(f, 0, true); if (rc) return rc;
  }
  f->first_done = true;
  return IA3_OK;
}

// one synchronisation for everything the caller wants back
int ia3_fit_results_ex(ia3_fitter* f, float* ps, uint8_t* success, int* nvox, int* n_iter) {
  if (!f) return set_error(IA3_EINVAL, "null fitter");
  hipStream_t st = stream();
  std::vector<SeedState> stv;
  if (n_iter) *n_iter = 0;
  StageCtl hc;
  memset(&hc, 0, sizeof(hc));
  int ovf = 0;
  if (f->n > 0) {
    // one copy: the four 256-byte control slots and the row table sit back to back in the pool (already on the host
    // when the last sweep check found nothing left to refit)
    const size_t head = (size_t)((char*)f->d_ps - (char*)f->d_counters);
    const size_t rows = ps ? sizeof(float) * 11 * (size_t)f->n : 0;
    std::vector<char>& hb = f->host_stage;
    const bool have = f->cached && hb.size() >= head + rows;
    // the fit may still be running: wait for it BEFORE queueing copies (see fetch_block: a copy queued behind a long
    // kernel holds a copy engine's ring for every other stream)
    if (!have || nvox || success) { const int rcw = stream_wait_spin(st); if (rcw) return rcw; }
    if (!have) {
      hb.resize(head + rows);
      IA3_HIP(hipMemcpyAsync(hb.data(), f->d_counters, head + rows, hipMemcpyDeviceToHost, st));
    }
    if (nvox) IA3_HIP(hipMemcpyAsync(nvox, f->d_nvox, sizeof(int) * (size_t)f->n, hipMemcpyDeviceToHost, st));
    if (success) {
      stv.resize(f->n);
      IA3_HIP(hipMemcpyAsync(stv.data(), f->d_state, sizeof(SeedState) * (size_t)f->n, hipMemcpyDeviceToHost, st));
    }
    if (!have || nvox || success) IA3_HIP(hipStreamSynchronize(st));
    memcpy(f->host_counters, hb.data(), sizeof(f->host_counters));
    if (n_iter) {   // the sweeps of the field that needed most (one field: its n_iter)
      const int* it = (const int*)(hb.data() + ((char*)f->d_niter - (char*)f->d_counters));
      for (size_t k = 0; k < f->ims.size(); ++k) if (it[k] > *n_iter) *n_iter = it[k];
    }
    memcpy(&hc, hb.data() + ((char*)f->d_ctl - (char*)f->d_counters), sizeof(StageCtl));
    memcpy(&ovf, hb.data() + ((char*)f->d_nbr_overflow - (char*)f->d_counters), sizeof(int));
    if (rows) memcpy(ps, hb.data() + head, rows);
  }
  { const int rc_ = check_ctl(hc); if (rc_) return rc_; }
  (void)ovf;   // > MAXNB neighbours somewhere: those seeds scanned the seed list instead (each_neighbour); not an error
  if (success) for (int i = 0; i < f->n; ++i) success[i] = (uint8_t)stv[i].success;
  return IA3_OK;
}

int ia3_fit_results(ia3_fitter* f, float* ps, uint8_t* success, int* nvox) {
  return ia3_fit_results_ex(f, ps, success, nvox, nullptr);
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

int ia3_fit_counters(ia3_fitter* f, uint64_t* out32) {
  if (!f || !out32) return set_error(IA3_EINVAL, "null argument");
  IA3_HIP(hipStreamSynchronize(stream()));
  IA3_HIP(hipMemcpy(out32, f->d_counters, 32 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return IA3_OK;
}

int ia3_fit_nfev(ia3_fitter* f, int* nfev) {
  if (!f || !nfev) return set_error(IA3_EINVAL, "null argument");
  IA3_HIP(hipStreamSynchronize(stream()));
  IA3_HIP(hipMemcpy(nfev, f->d_nfev, sizeof(int) * (size_t)f->n, hipMemcpyDeviceToHost));
  return IA3_OK;
}

int ia3_fit_seeds(const void* im, int dtype, int Z, int X, int Y, const double* centers_zxy, int n,
                  const ia3_fit_params* p, float* out_ps, uint8_t* success, int* n_iter) {
  ia3_stack* s = nullptr;
  ia3_fitter* f = nullptr;
  int rc = ia3_stack_upload(im, dtype, Z, X, Y, &s); if (rc) return rc;
  rc = ia3_fit_create(s, centers_zxy, n, p, &f);
  if (!rc) rc = ia3_fit_run(f);
  if (!rc) rc = ia3_fit_results_ex(f, out_ps, success, nullptr, n_iter);
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
