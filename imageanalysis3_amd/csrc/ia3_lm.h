// Levenberg–Marquardt driver with MINPACK lmder's control logic (Moré 1978; the algorithm
// scipy.optimize.leastsq runs for the reference, External/Fitting_v4.py:388) restated on the
// normal equations:  A = JᵀJ, g = Jᵀf, |f|.
//
// lmder needs from its QR factorisation only quantities that are functions of A and g:
//   column norms            acnorm_j = sqrt(A_jj)
//   scaled gradient norm    gnorm    = max_j |g_j| / (|f| acnorm_j)
//   LM step p(par)          (A + par D²) p = -g          (lmpar/qrsolv solve exactly this system)
//   newton correction       |S⁻ᵀ D q|² = qᵀ D (A + par D²)⁻¹ D q
//   predicted reduction     |J p|²   = pᵀ A p
// so the trust-region iteration (delta/par updates, ratio test, the ftol/xtol/gtol stopping
// rules, maxfev, factor=100, mode=1 scaling) is reproduced step for step, and the only
// numerical difference to MINPACK is Cholesky on A instead of Householder QR on J (both f64).
//
// Exactly-zero Jacobian columns (the two angle columns at the start point, where all three
// widths are equal) are handled the way lmpar's rank-deficient branch does: their step
// component is zero and the Newton lower bound parl is not used.
//
// Every loop has compile-time bounds (NP = 10) and is fully unrolled, so on the GPU all matrix
// and vector accesses are statically indexed: the persistent state (LMWork, in LDS) is read with
// immediate offsets and the per-call temporaries (Cholesky factor, work vectors) live in
// registers.  The factor stores reciprocal diagonals: ten reciprocal square roots per factorisation (lm_rsqrt: the
// hardware estimate plus two Newton steps on the device, 1 / sqrt on the host — the only place where the two builds
// differ, by about an ulp).
//
// Host/device agnostic. `Eval` supplies:  double eval(const double* x, double* A, double* g)
// which returns |f(x)| and fills the packed upper triangle of JᵀJ (row-major, NTRI entries) and
// g = Jᵀf at the same x.
#pragma once
#include "ia3_model.h"

namespace ia3 {

IA3_HD constexpr int tri(int i, int j) {  // packed upper-triangle index, i <= j
  return i * NP - (i * (i - 1)) / 2 + (j - i);
}

#if defined(__HIPCC__) || defined(__clang__)
#define IA3_UNROLL _Pragma("unroll")
#define IA3_NOUNROLL _Pragma("clang loop unroll(disable)")
#else
#define IA3_NOUNROLL
#define IA3_UNROLL _Pragma("GCC unroll 16")
#endif

// Element-wise sections (ten independent square roots or divisions, each a 20-30 instruction sequence on the GPU) run
// on lanes 0..9 of the wave, one element per lane, instead of ten times on every lane: operands and results go through
// the LDS-resident work area, the surrounding control flow stays wave-uniform.  Same operations on the same operands,
// so results are bit-identical to the serial form the host build uses.
#if defined(__HIP_DEVICE_COMPILE__)
#define IA3_LM_PAR 1
#define IA3_LM_SYNC() __builtin_amdgcn_wave_barrier()
// lane index inside a group of LW lanes (64: one fit per wave; 32: the paired solver of fit.hip, one fit per half-wave)
template <int LW> __device__ __forceinline__ int lm_lane() {
  return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) & (LW - 1);
}
#else
#define IA3_LM_PAR 0
#endif

struct LMWork {      // persistent across evaluations (LDS on the device)
  double A[NTRI];    // JᵀJ at the current accepted point
  double g[NP];      // Jᵀf at the current accepted point
  double A1[NTRI];   // the same at the trial point (kept if the step is accepted)
  double g1[NP];
  double diag[NP];
  double cn[NP];     // column norms sqrt(A_jj) of the current accepted point
  double x[NP];
  double xt[NP];     // trial point
  double p[NP];      // step
  double sc[2 * NP]; // scratch of the lane-parallel sections
};

struct LMResult {
  int info;
  int nfev;
  int iter;
  double fnorm;
};

#define IA3_EPSMCH 2.220446049250313e-16
#define IA3_DWARF 2.2250738585072014e-308

struct Chol {        // A + par D² = LᵀL, L upper; transient (registers)
  double L[NTRI];    // off-diagonals; L[tri(j,j)] holds 1 / L_jj (0 for a skipped column)
  unsigned skip;     // bit j: non-positive pivot, column treated as absent
};

// 1 / sqrt(s), s > 0.  The ten pivots of a factorisation are a dependent chain and a fit owns its SIMD (one wave):
// of the 4.3 k cycles lm_factor took, 3.6 k were the library's sqrt followed by a division, ten times in a row
// (scripts/lone_fit.py).  The hardware estimate refined by two Newton steps (error ~1 ulp) is a fifth of that; fits
// that converge are unaffected at float32 precision, the ones that stop at maxfev — whose end point depends on the
// last bit of every step anyway — take 8-15 % less time.
IA3_HD double lm_rsqrt(double s) {
#if defined(__HIP_DEVICE_COMPILE__)
  // (safe for every normal s: y <= 6.7e153, s * y = sqrt(s); lm_factor does not pass subnormal pivots)
  double y = __builtin_amdgcn_rsq(s);
  double e = __builtin_fma(-(s * y), y, 1.0);
  y = __builtin_fma(0.5 * y, e, y);
  e = __builtin_fma(-(s * y), y, 1.0);
  y = __builtin_fma(0.5 * y, e, y);
  return y;
#else
  return 1.0 / sqrt(s);
#endif
}

IA3_HD void lm_factor(const double* A, const double* diag, double par, Chol& c) {
  c.skip = 0;
  IA3_UNROLL
  for (int j = 0; j < NP; ++j) {
    IA3_UNROLL
    for (int i = 0; i <= j; ++i) {
      double s = A[tri(i, j)];
      if (i == j) s += par * diag[j] * diag[j];
      IA3_UNROLL
      for (int k = 0; k < i; ++k) s -= c.L[tri(k, i)] * c.L[tri(k, j)];
      if (i == j) {
        if (s >= IA3_DWARF) c.L[tri(j, j)] = lm_rsqrt(s);   // a subnormal pivot is a vanished column, like s <= 0
        else { c.L[tri(j, j)] = 0.0; c.skip |= 1u << j; }
      } else {
        c.L[tri(i, j)] = s * c.L[tri(i, i)];
      }
    }
  }
}
// y := L⁻ᵀ b  (forward substitution with Lᵀ), skipped components = 0; in-place safe
IA3_HD void lm_fwd(const Chol& c, const double* b, double* y) {
  IA3_UNROLL
  for (int j = 0; j < NP; ++j) {
    double s = b[j];
    IA3_UNROLL
    for (int k = 0; k < j; ++k) s -= c.L[tri(k, j)] * y[k];
    y[j] = s * c.L[tri(j, j)];
  }
}
// x := L⁻¹ y  (back substitution), skipped components = 0
IA3_HD void lm_bwd(const Chol& c, const double* y, double* x) {
  IA3_UNROLL
  for (int j = NP - 1; j >= 0; --j) {
    double s = y[j];
    IA3_UNROLL
    for (int k = j + 1; k < NP; ++k) s -= c.L[tri(j, k)] * x[k];
    x[j] = s * c.L[tri(j, j)];
  }
}
IA3_HD double lm_norm(const double* v) {
  double s = 0;
  IA3_UNROLL
  for (int j = 0; j < NP; ++j) s += v[j] * v[j];
  return sqrt(s);
}

// t1[j] = diag[j] * ((diag[j] * x[j]) / dxnorm);  t2[j] = diag[j] * x[j] is passed in for the serial form
template <int LW = 64>
IA3_HD void lm_scaled(const double* diag, const double* x, const double* t2, double dxnorm, double* t1, double* sc) {
#if IA3_LM_PAR
  IA3_UNROLL
  for (int j = 0; j < NP; ++j) sc[NP + j] = x[j];          // wave-uniform values: every lane stores the same
  IA3_LM_SYNC();
  const int ln = lm_lane<LW>();
  if (ln < NP) { const double d = diag[ln]; sc[ln] = d * ((d * sc[NP + ln]) / dxnorm); }
  IA3_LM_SYNC();
  IA3_UNROLL
  for (int j = 0; j < NP; ++j) t1[j] = sc[j];
  IA3_LM_SYNC();
  (void)t2;
#else
  IA3_UNROLL
  for (int j = 0; j < NP; ++j) t1[j] = diag[j] * (t2[j] / dxnorm);
  (void)x; (void)sc;
#endif
}

// lmpar on the normal equations: find par with | |D x| - delta | <= 0.1 delta, x = (A+par D²)⁻¹ g
// sc: 2*NP doubles of scratch (LDS on the device)
// One loop, one factorisation site: pass 0 is lmpar's Gauss–Newton trial (par = 0) with its bounds parl / paru,
// passes 1..10 its Newton iteration on par — the same operations in the same order as the two-part form MINPACK
// writes, in half the code (the fit kernel is instruction-cache bound).
template <int LW = 64>
IA3_HD void lm_par(const double* A, const double* g, const double* diag, double delta, double& par, double* x,
                   double* sc) {
  Chol c;
  double t1[NP], t2[NP];
  double parl = 0.0, paru = 0.0, fp = 0.0;
  IA3_NOUNROLL
  for (int iter = 0;; ++iter) {
    double pf = 0.0;
    if (iter > 0) {
      if (par == 0.0) { double t = 0.001 * paru; par = IA3_DWARF > t ? IA3_DWARF : t; }
      pf = par;
    }
    lm_factor(A, diag, pf, c);
    lm_fwd(c, g, t1);
    lm_bwd(c, t1, x);
    IA3_UNROLL
    for (int j = 0; j < NP; ++j) t2[j] = diag[j] * x[j];
    const double dxnorm = lm_norm(t2);
    const double fp_old = fp;
    fp = dxnorm - delta;
    if (iter == 0) {
      if (fp <= 0.1 * delta) { par = 0.0; return; }
    } else if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= fp_old && fp_old < 0.0) || iter == 10) {
      return;
    }
    double parc = 0.0;   // Newton correction; pass 0: lmpar's lower bound, not used when a column vanished
    if (iter > 0 || c.skip == 0) {
      lm_scaled<LW>(diag, x, t2, dxnorm, t1, sc);
      lm_fwd(c, t1, t1);
      const double temp = lm_norm(t1);
      parc = ((fp / delta) / temp) / temp;
    }
    if (iter == 0) {
      parl = parc;
#if IA3_LM_PAR
      {
        const int ln = lm_lane<LW>();
        if (ln < NP) sc[ln] = g[ln] / diag[ln];
        IA3_LM_SYNC();
        IA3_UNROLL
        for (int j = 0; j < NP; ++j) t1[j] = sc[j];
        IA3_LM_SYNC();
      }
#else
      IA3_UNROLL
      for (int j = 0; j < NP; ++j) t1[j] = g[j] / diag[j];
#endif
      const double gnorm = lm_norm(t1);
      paru = gnorm / delta;
      if (paru == 0.0) paru = IA3_DWARF / (delta < 0.1 ? delta : 0.1);
      par = par > parl ? par : parl;
      par = par < paru ? par : paru;
      if (par == 0.0) par = gnorm / dxnorm;
    } else {
      if (fp > 0.0) parl = parl > par ? parl : par;
      if (fp < 0.0) paru = paru < par ? paru : par;
      const double pn = par + parc;
      par = parl > pn ? parl : pn;
    }
  }
}

// lmder's iteration with ONE evaluation site: the evaluation at the start point is the first pass of the same loop
// that evaluates the trial points (it is "accepted" unconditionally).  After an accepted point the outer-loop
// prologue of lmder runs (column norms, scaling, gradient test), then lmpar and the next trial point.
template <class Eval>
IA3_HD LMResult lm_solve(Eval& ev, LMWork& w, double ftol, double xtol, double gtol, int maxfev,
                         double factor) {
  LMResult r;
  r.info = 0; r.nfev = 0; r.iter = 1;
  // JᵀJ / Jᵀf of the accepted point and of the trial point ping-pong between the two halves of the work area:
  // accepting a step swaps the pointers instead of copying 65 values through LDS
  double* Ac = w.A; double* gc = w.g;      // current accepted point
  double* At = w.A1; double* gt = w.g1;    // trial point
  double fnorm = 0.0, par = 0.0, delta = 0.0, xnorm = 0.0, gnorm = 0.0, pnorm = 0.0, jp2 = 0.0;
  bool first = true;
  IA3_NOUNROLL
  for (;;) {
    const double fnorm1 = ev.eval(first ? w.x : w.xt, At, gt);
    ++r.nfev;
    bool accepted;
    if (first) {
      first = false;
      accepted = true;
      fnorm = fnorm1;
      { double* sw = Ac; Ac = At; At = sw; sw = gc; gc = gt; gt = sw; }
    } else {
      double actred = -1.0;
      if (0.1 * fnorm1 < fnorm) { double q = fnorm1 / fnorm; actred = 1.0 - q * q; }
      double temp1 = sqrt(jp2) / fnorm;
      double temp2 = (sqrt(par) * pnorm) / fnorm;
      double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
      double dirder = -(temp1 * temp1 + temp2 * temp2);
      double ratio = prered != 0.0 ? actred / prered : 0.0;
      if (ratio <= 0.25) {
        double temp;
        if (actred >= 0.0) temp = 0.5;
        else temp = 0.5 * dirder / (dirder + 0.5 * actred);
        if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
        double pm = pnorm / 0.1;
        delta = temp * (delta < pm ? delta : pm);
        par = par / temp;
      } else if (par == 0.0 || ratio >= 0.75) {
        delta = pnorm / 0.5;
        par = 0.5 * par;
      }
      accepted = ratio >= 1e-4;
      if (accepted) {  // successful iteration
        double t[NP];
        IA3_UNROLL
        for (int j = 0; j < NP; ++j) { w.x[j] = w.xt[j]; t[j] = w.diag[j] * w.xt[j]; }
        { double* sw = Ac; Ac = At; At = sw; sw = gc; gc = gt; gt = sw; }
        xnorm = lm_norm(t);
        fnorm = fnorm1;
        ++r.iter;
      }
      bool small = fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0;
      if (small) r.info = 1;
      if (delta <= xtol * xnorm) r.info = 2;
      if (small && r.info == 2) r.info = 3;
      if (r.info != 0) break;
      if (r.nfev >= maxfev) r.info = 5;
      if (fabs(actred) <= IA3_EPSMCH && prered <= IA3_EPSMCH && 0.5 * ratio <= 1.0) r.info = 6;
      if (delta <= IA3_EPSMCH * xnorm) r.info = 7;
      if (gnorm <= IA3_EPSMCH) r.info = 8;
      if (r.info != 0) break;
    }
    if (accepted) {  // lmder's outer loop: Ac, gc hold JᵀJ, Jᵀf at x
#if IA3_LM_PAR
      {
        const int ln = lm_lane<64>();
        if (ln < NP) w.cn[ln] = sqrt(Ac[tri(ln, ln)]);
        IA3_LM_SYNC();
      }
#else
      IA3_UNROLL
      for (int j = 0; j < NP; ++j) w.cn[j] = sqrt(Ac[tri(j, j)]);
#endif
      if (r.iter == 1) {
        double t[NP];
        IA3_UNROLL
        for (int j = 0; j < NP; ++j) {
          w.diag[j] = w.cn[j] == 0.0 ? 1.0 : w.cn[j];
          t[j] = w.diag[j] * w.x[j];
        }
        xnorm = lm_norm(t);
        delta = factor * xnorm;
        if (delta == 0.0) delta = factor;
      }
      gnorm = 0.0;
      if (fnorm != 0.0) {
#if IA3_LM_PAR
        {
          const int ln = lm_lane<64>();
          if (ln < NP) { const double cnl = w.cn[ln]; if (cnl != 0.0) w.sc[ln] = fabs((gc[ln] / fnorm) / cnl); }
          IA3_LM_SYNC();
        }
#endif
        IA3_UNROLL
        for (int j = 0; j < NP; ++j) {
          if (w.cn[j] != 0.0) {
#if IA3_LM_PAR
            double v = w.sc[j];
#else
            double v = fabs((gc[j] / fnorm) / w.cn[j]);
#endif
            gnorm = gnorm > v ? gnorm : v;
          }
        }
#if IA3_LM_PAR
        IA3_LM_SYNC();
#endif
      }
      if (gnorm <= gtol) { r.info = 4; break; }
#if IA3_LM_PAR
      {
        const int ln = lm_lane<64>();
        if (ln < NP) { const double dl = w.diag[ln], cl = w.cn[ln]; w.diag[ln] = dl > cl ? dl : cl; }
        IA3_LM_SYNC();
      }
#else
      IA3_UNROLL
      for (int j = 0; j < NP; ++j) w.diag[j] = w.diag[j] > w.cn[j] ? w.diag[j] : w.cn[j];
#endif
    }
    {  // lmder's inner loop up to the evaluation of the trial point
      double pv[NP], t[NP];
      lm_par(Ac, gc, w.diag, delta, par, pv, w.sc);
      IA3_UNROLL
      for (int j = 0; j < NP; ++j) {
        pv[j] = -pv[j];
        w.p[j] = pv[j];
        w.xt[j] = w.x[j] + pv[j];
        t[j] = w.diag[j] * pv[j];
      }
      pnorm = lm_norm(t);
      if (r.iter == 1) delta = delta < pnorm ? delta : pnorm;
      // |J p|² = pᵀ A p   (A of the current accepted point: the evaluation writes At / gt)
      jp2 = 0.0;
#if IA3_LM_PAR
      {
        const int ln = lm_lane<64>();
        if (ln < NP) {   // row ln of A times p, columns in ascending order as in the serial form
          double s = 0.0;
          IA3_UNROLL
          for (int j = 0; j < NP; ++j) s += (ln <= j ? Ac[tri(ln, j)] : Ac[tri(j, ln)]) * pv[j];
          w.sc[ln] = s;
        }
        IA3_LM_SYNC();
        IA3_UNROLL
        for (int i = 0; i < NP; ++i) jp2 += pv[i] * w.sc[i];
        IA3_LM_SYNC();
      }
#else
      IA3_UNROLL
      for (int i = 0; i < NP; ++i) {
        double s = 0.0;
        IA3_UNROLL
        for (int j = 0; j < NP; ++j) s += (i <= j ? Ac[tri(i, j)] : Ac[tri(j, i)]) * pv[j];
        jp2 += pv[i] * s;
      }
#endif
      if (jp2 < 0.0) jp2 = 0.0;
    }
  }
  r.fnorm = fnorm;
  return r;
}

}  // namespace ia3
