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
// Host/device agnostic. `Eval` supplies:  double eval(const double* x, double* A, double* g)
// which returns |f(x)| and, when A != nullptr, fills the packed upper triangle of JᵀJ (row-major,
// NTRI entries) and g = Jᵀf at the same x.
#pragma once
#include "ia3_model.h"

namespace ia3 {

IA3_HD int tri(int i, int j) {  // packed upper-triangle index, i <= j
  return i * NP - (i * (i - 1)) / 2 + (j - i);
}

struct LMWork {
  double A[NTRI];    // JᵀJ at the current accepted point
  double g[NP];      // Jᵀf at the current accepted point
  double A1[NTRI];   // the same at the trial point (kept if the step is accepted)
  double g1[NP];
  double L[NTRI];    // Cholesky factor (upper, packed): A + par D² = LᵀL
  double diag[NP];
  double x[NP];
  double xt[NP];     // trial point
  double p[NP];      // step
  double w1[NP];
  double w2[NP];
  int skip[NP];      // 1 = zero pivot (column treated as absent)
};

struct LMResult {
  int info;
  int nfev;
  int iter;
  double fnorm;
};

#define IA3_EPSMCH 2.220446049250313e-16
#define IA3_DWARF 2.2250738585072014e-308

// Cholesky of M = A + par*diag² (upper packed). Non-positive pivots -> column skipped.
IA3_HD int lm_factor(const double* A, const double* diag, double par, double* L, int* skip) {
  int nskip = 0;
  for (int j = 0; j < NP; ++j) {
    for (int i = 0; i <= j; ++i) {
      double s = A[tri(i, j)];
      if (i == j) s += par * diag[j] * diag[j];
      for (int k = 0; k < i; ++k) s -= L[tri(k, i)] * L[tri(k, j)];
      if (i == j) {
        if (s > 0.0) { L[tri(j, j)] = sqrt(s); skip[j] = 0; }
        else { L[tri(j, j)] = 0.0; skip[j] = 1; ++nskip; }
      } else {
        L[tri(i, j)] = skip[i] ? 0.0 : s / L[tri(i, i)];
      }
    }
  }
  return nskip;
}
// y := L⁻ᵀ b  (forward substitution with Lᵀ), skipped components = 0
IA3_HD void lm_fwd(const double* L, const int* skip, const double* b, double* y) {
  for (int j = 0; j < NP; ++j) {
    double s = b[j];
    for (int k = 0; k < j; ++k) s -= L[tri(k, j)] * y[k];
    y[j] = skip[j] ? 0.0 : s / L[tri(j, j)];
  }
}
// x := L⁻¹ y  (back substitution), skipped components = 0
IA3_HD void lm_bwd(const double* L, const int* skip, const double* y, double* x) {
  for (int j = NP - 1; j >= 0; --j) {
    double s = y[j];
    for (int k = j + 1; k < NP; ++k) s -= L[tri(j, k)] * x[k];
    x[j] = skip[j] ? 0.0 : s / L[tri(j, j)];
  }
}
IA3_HD double lm_norm(const double* v) {
  double s = 0;
  for (int j = 0; j < NP; ++j) s += v[j] * v[j];
  return sqrt(s);
}

// lmpar on the normal equations: find par with | |D x| - delta | <= 0.1 delta, x = (A+par D²)⁻¹ g
IA3_HD void lm_par(LMWork& w, double delta, double& par, double* x) {
  int nskip = lm_factor(w.A, w.diag, 0.0, w.L, w.skip);
  lm_fwd(w.L, w.skip, w.g, w.w1);
  lm_bwd(w.L, w.skip, w.w1, x);
  int iter = 0;
  for (int j = 0; j < NP; ++j) w.w2[j] = w.diag[j] * x[j];
  double dxnorm = lm_norm(w.w2);
  double fp = dxnorm - delta;
  if (fp <= 0.1 * delta) { par = 0.0; return; }
  double parl = 0.0;
  if (nskip == 0) {
    for (int j = 0; j < NP; ++j) w.w1[j] = w.diag[j] * (w.w2[j] / dxnorm);
    lm_fwd(w.L, w.skip, w.w1, w.w1);
    double temp = lm_norm(w.w1);
    parl = ((fp / delta) / temp) / temp;
  }
  for (int j = 0; j < NP; ++j) w.w1[j] = w.g[j] / w.diag[j];
  double gnorm = lm_norm(w.w1);
  double paru = gnorm / delta;
  if (paru == 0.0) paru = IA3_DWARF / (delta < 0.1 ? delta : 0.1);
  par = par > parl ? par : parl;
  par = par < paru ? par : paru;
  if (par == 0.0) par = gnorm / dxnorm;
  for (;;) {
    ++iter;
    if (par == 0.0) { double t = 0.001 * paru; par = IA3_DWARF > t ? IA3_DWARF : t; }
    lm_factor(w.A, w.diag, par, w.L, w.skip);
    lm_fwd(w.L, w.skip, w.g, w.w1);
    lm_bwd(w.L, w.skip, w.w1, x);
    for (int j = 0; j < NP; ++j) w.w2[j] = w.diag[j] * x[j];
    dxnorm = lm_norm(w.w2);
    double temp = fp;
    fp = dxnorm - delta;
    if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || iter == 10) break;
    for (int j = 0; j < NP; ++j) w.w1[j] = w.diag[j] * (w.w2[j] / dxnorm);
    lm_fwd(w.L, w.skip, w.w1, w.w1);
    temp = lm_norm(w.w1);
    double parc = ((fp / delta) / temp) / temp;
    if (fp > 0.0) parl = parl > par ? parl : par;
    if (fp < 0.0) paru = paru < par ? paru : par;
    double pn = par + parc;
    par = parl > pn ? parl : pn;
  }
}

template <class Eval>
IA3_HD LMResult lm_solve(Eval& ev, LMWork& w, double ftol, double xtol, double gtol, int maxfev,
                         double factor) {
  LMResult r;
  r.info = 0; r.nfev = 1; r.iter = 1;
  double fnorm = ev.eval(w.x, w.A, w.g);
  double par = 0.0, delta = 0.0, xnorm = 0.0;
  for (;;) {  // outer loop: A, g hold JᵀJ, Jᵀf at x
    if (r.iter == 1) {
      for (int j = 0; j < NP; ++j) {
        double cn = sqrt(w.A[tri(j, j)]);
        w.diag[j] = cn == 0.0 ? 1.0 : cn;
        w.w1[j] = w.diag[j] * w.x[j];
      }
      xnorm = lm_norm(w.w1);
      delta = factor * xnorm;
      if (delta == 0.0) delta = factor;
    }
    double gnorm = 0.0;
    if (fnorm != 0.0) {
      for (int j = 0; j < NP; ++j) {
        double cn = sqrt(w.A[tri(j, j)]);
        if (cn != 0.0) {
          double v = fabs((w.g[j] / fnorm) / cn);
          gnorm = gnorm > v ? gnorm : v;
        }
      }
    }
    if (gnorm <= gtol) { r.info = 4; break; }
    for (int j = 0; j < NP; ++j) {
      double cn = sqrt(w.A[tri(j, j)]);
      w.diag[j] = w.diag[j] > cn ? w.diag[j] : cn;
    }
    for (;;) {  // inner loop
      lm_par(w, delta, par, w.p);
      for (int j = 0; j < NP; ++j) {
        w.p[j] = -w.p[j];
        w.xt[j] = w.x[j] + w.p[j];
        w.w1[j] = w.diag[j] * w.p[j];
      }
      double pnorm = lm_norm(w.w1);
      if (r.iter == 1) delta = delta < pnorm ? delta : pnorm;
      double fnorm1 = ev.eval(w.xt, w.A1, w.g1);
      ++r.nfev;
      double actred = -1.0;
      if (0.1 * fnorm1 < fnorm) { double q = fnorm1 / fnorm; actred = 1.0 - q * q; }
      // |J p|² = pᵀ A p
      double jp2 = 0.0;
      for (int i = 0; i < NP; ++i) {
        double s = 0.0;
        for (int j = 0; j < NP; ++j) s += (i <= j ? w.A[tri(i, j)] : w.A[tri(j, i)]) * w.p[j];
        jp2 += w.p[i] * s;
      }
      if (jp2 < 0.0) jp2 = 0.0;
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
      if (ratio >= 1e-4) {  // successful iteration
        for (int j = 0; j < NP; ++j) { w.x[j] = w.xt[j]; w.w1[j] = w.diag[j] * w.x[j]; w.g[j] = w.g1[j]; }
        for (int k = 0; k < NTRI; ++k) w.A[k] = w.A1[k];
        xnorm = lm_norm(w.w1);
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
      if (ratio >= 1e-4) break;  // leave inner loop, new Jacobian already in A, g
    }
    if (r.info != 0) break;
  }
  r.fnorm = fnorm;
  return r;
}

}  // namespace ia3
