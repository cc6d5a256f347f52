// 10-parameter sigmoid-constrained 3-D Gaussian model: geometry set-up, residual and Jacobian row.
//
// Follows the arithmetic of the reference's GaussianFit (External/Fitting_v4.py:189-375):
// float64 evaluation of f = exp(bk) + exp(h - xᵀAx/2) from float32 voxel values/coordinates,
// analytic Jacobian evaluated in float64 and rounded to float32 per entry (:365).
// Parameter order of the unconstrained vector: [bk, h, c0, c1, c2, w0, w1, w2, pp, tp]
// (the reference's "x,y,z" are array axes 0,1,2 = z,x,y of the stack).
//
// Host/device agnostic: compiled by hipcc for the wave-per-fit kernel (fit.hip) and by g++
// for the CPU unit tests of the solver (tests/test_lm_core_cpu.py via lm_cpu.cpp).
#pragma once
#include <math.h>
#include <float.h>

#if defined(__HIPCC__)
#define IA3_HD __host__ __device__ __forceinline__
#else
#define IA3_HD inline
#endif

namespace ia3 {

constexpr int NP = 10;                 // free parameters
constexpr int NTRI = NP * (NP + 1) / 2;  // packed upper triangle of JᵀJ

struct FitCfg {
  double min_ws;   // min_w²
  double max_ws;   // max_w²
  double delta;    // delta_center
  double init_w;
  double c0[3];    // centre estimate (the seed)
  // model variant: 0 = External/Fitting_v4.py (production); 1 = External/Fitting_v3.py (legacy
  // `_fit_single_image` path, classes/__init__.py:57-88): per-axis start widths (Fitting_v3.py:71-79) and its
  // to_center (:81-87), whose third coordinate mixes the second and third offsets.
  int variant;     // set it: no default member initialisers, the struct also lives in LDS
  double iw[3];    // variant 1: start value of the three width parameters (already in w_ space)
};

// log(DBL_MAX): the reference's overflow guard (np.log(np.finfo(float64).max)); parameters
// arriving from the solver are float64 in the reference, so this limit applies on the whole path.
#define IA3_LOGMAX 709.782712893384

IA3_HD double sig_center(double c_, double delta, double c0) {
  if (c_ >= IA3_LOGMAX) return -delta + c0;
  if (c_ <= -IA3_LOGMAX) return delta + c0;
  return 2. * delta / (1. + exp(c_)) - delta + c0;
}
// Fitting_v3.py:84-86: 2*delta*exp(-a)/(1+exp(-b)) - delta + c0 (a == b for the first two axes)
IA3_HD double sig_center_v3(double a, double b, double delta, double c0) {
  return 2. * delta * exp(-a) / (1. + exp(-b)) - delta + c0;
}
IA3_HD void centers_of(const double* x, const FitCfg& cfg, double* c) {
  if (cfg.variant == 1) {
    c[0] = sig_center_v3(x[2], x[2], cfg.delta, cfg.c0[0]);
    c[1] = sig_center_v3(x[3], x[3], cfg.delta, cfg.c0[1]);
    c[2] = sig_center_v3(x[3], x[4], cfg.delta, cfg.c0[2]);
  } else {
    c[0] = sig_center(x[2], cfg.delta, cfg.c0[0]);
    c[1] = sig_center(x[3], cfg.delta, cfg.c0[1]);
    c[2] = sig_center(x[4], cfg.delta, cfg.c0[2]);
  }
}
IA3_HD double sig_sine(double t_) {
  if (t_ >= IA3_LOGMAX) return -1.;
  if (t_ <= -IA3_LOGMAX) return 1.;
  return 2. / (1. + exp(t_)) - 1.;
}
IA3_HD double sig_ws(double w_, double min_ws, double max_ws) {
  double dws = max_ws - min_ws;
  if (w_ >= IA3_LOGMAX) return min_ws;
  if (w_ <= -IA3_LOGMAX) return dws + min_ws;
  return dws / (1. + exp(w_)) + min_ws;
}
IA3_HD double norm_w(double w, double minw, double maxw) {
  if (w > 0) {
    double e = exp(-w);
    double d = maxw * e + minw;
    return 0.5 * (maxw - minw) * e / (d * d);
  }
  double e = exp(w);
  double d = minw * e + maxw;
  return 0.5 * (maxw - minw) * e / (d * d);
}

// Everything that depends on the parameter vector only (wave-uniform on the GPU).
struct Geom {
  double h, ebk_f, ebk_j;   // exp(clip(bk)) for f ; exp(bk) for the Jacobian column (:287 vs :347)
  double c[3];              // centre
  double q[6];              // x2c, y2c, z2c, xyc, xzc, yzc
  // Jacobian columns 2..4: linear forms  F * (l[k][0]*xt + l[k][1]*yt + l[k][2]*zt)
  double l[3][3];
  // Jacobian columns 5..9: quadratic forms F * (m[k] · [xt², xt·yt, yt², xt·zt, yt·zt, zt²])
  double m[5][6];
};

// The transcendental part of the geometry: everything that needs an exponential, a division or a square root.
struct GeomScalars {
  double t, p, tc, pc;        // sines of the two angles and their cosines
  double s[3];                // 1 / ws_k
  double c[3];                // centre
  double nc[3];               // d centre_k / d (unconstrained centre_k)
  double nw[3];               // d ws_k / d w_k (norm_w)
  double np_, nt_;            // derivative factors of the two angle sigmoids
  double ebk_f, ebk_j;
};

IA3_HD void geom_scalars(const double* x, const FitCfg& cfg, GeomScalars& q) {
  const double bk = x[0], xp = x[2], yp = x[3], zp = x[4];
  const double w1 = x[5], w2 = x[6], w3 = x[7], pp = x[8], tp = x[9];
  q.t = sig_sine(tp); q.p = sig_sine(pp);
  const double ws1 = sig_ws(w1, cfg.min_ws, cfg.max_ws);
  const double ws2 = sig_ws(w2, cfg.min_ws, cfg.max_ws);
  const double ws3 = sig_ws(w3, cfg.min_ws, cfg.max_ws);
  centers_of(x, cfg, q.c);
  const double p2 = q.p * q.p, t2 = q.t * q.t, tc2 = 1 - t2, pc2 = 1 - p2;
  q.tc = sqrt(tc2); q.pc = sqrt(pc2);
  q.s[0] = 1. / ws1; q.s[1] = 1. / ws2; q.s[2] = 1. / ws3;
  double bkc = bk < -709.78 ? -709.78 : (bk > 709.78 ? 709.78 : bk);
  q.ebk_f = cfg.variant == 1 ? exp(bk) : exp(bkc);   // Fitting_v3.py:119 has no clip
  q.ebk_j = exp(bk);
  const double d = cfg.delta;
  const double e_xp = exp(-fabs(xp)), e_yp = exp(-fabs(yp)), e_zp = exp(-fabs(zp));
  q.nc[0] = -d * e_xp / ((1 + e_xp) * (1 + e_xp));
  q.nc[1] = -d * e_yp / ((1 + e_yp) * (1 + e_yp));
  q.nc[2] = -d * e_zp / ((1 + e_zp) * (1 + e_zp));
  q.nw[0] = norm_w(w1, cfg.min_ws, cfg.max_ws);
  q.nw[1] = norm_w(w2, cfg.min_ws, cfg.max_ws);
  q.nw[2] = norm_w(w3, cfg.min_ws, cfg.max_ws);
  const double e_p = exp(-fabs(pp) / 2), e_t = exp(-fabs(tp) / 2);
  q.np_ = e_p / (1 + e_p * e_p); q.nt_ = e_t / (1 + e_t * e_t);
}

// products only: the coefficient tables of f and of the Jacobian columns
IA3_HD void geom_assemble(const double* x, const GeomScalars& q, Geom& g) {
  const double h = x[1];
  const double t = q.t, p = q.p, tc = q.tc, pc = q.pc;
  const double p2 = p * p, t2 = t * t, tc2 = 1 - t2, pc2 = 1 - p2;
  const double s1 = q.s[0], s2 = q.s[1], s3 = q.s[2];
  g.c[0] = q.c[0]; g.c[1] = q.c[1]; g.c[2] = q.c[2];
  const double x2c = pc2 * tc2 * s1 + t2 * s2 + p2 * tc2 * s3;
  const double y2c = pc2 * t2 * s1 + tc2 * s2 + p2 * t2 * s3;
  const double z2c = p2 * s1 + pc2 * s3;
  const double xyc = 2 * tc * t * (pc2 * s1 - s2 + p2 * s3);
  const double xzc = 2 * p * pc * tc * (s3 - s1);
  const double yzc = 2 * p * pc * t * (s3 - s1);
  g.q[0] = x2c; g.q[1] = y2c; g.q[2] = z2c; g.q[3] = xyc; g.q[4] = xzc; g.q[5] = yzc;
  g.h = h;
  g.ebk_f = q.ebk_f;
  g.ebk_j = q.ebk_j;
  const double nxp = q.nc[0], nyp = q.nc[1], nzp = q.nc[2];
  g.l[0][0] = 2 * x2c * nxp; g.l[0][1] = xyc * nxp;     g.l[0][2] = xzc * nxp;
  g.l[1][0] = xyc * nyp;     g.l[1][1] = 2 * y2c * nyp; g.l[1][2] = yzc * nyp;
  g.l[2][0] = xzc * nzp;     g.l[2][1] = yzc * nzp;     g.l[2][2] = 2 * z2c * nzp;
  const double nw1 = q.nw[0], nw2 = q.nw[1], nw3 = q.nw[2];
  // order of the 6 monomials: xt², xt·yt, yt², xt·zt, yt·zt, zt²
  g.m[0][0] = -pc2 * tc2 * nw1;       g.m[0][1] = -2 * pc2 * t * tc * nw1; g.m[0][2] = -pc2 * t2 * nw1;
  g.m[0][3] = 2 * p * pc * tc * nw1;  g.m[0][4] = 2 * p * pc * t * nw1;    g.m[0][5] = -p2 * nw1;
  g.m[1][0] = -t2 * nw2;              g.m[1][1] = 2 * t * tc * nw2;        g.m[1][2] = -tc2 * nw2;
  g.m[1][3] = 0;                      g.m[1][4] = 0;                       g.m[1][5] = 0;
  g.m[2][0] = -p2 * tc2 * nw3;        g.m[2][1] = -2 * p2 * t * tc * nw3;  g.m[2][2] = -p2 * t2 * nw3;
  g.m[2][3] = -2 * p * pc * tc * nw3; g.m[2][4] = -2 * p * pc * t * nw3;   g.m[2][5] = -pc2 * nw3;
  const double np_ = q.np_, nt_ = q.nt_;
  const double a9 = (s3 - s1) * np_, b9 = 2 * pc2 - 1., ppc = p * pc;
  g.m[3][0] = a9 * ppc * tc2;         g.m[3][1] = a9 * ppc * 2 * t * tc;   g.m[3][2] = a9 * ppc * t2;
  g.m[3][3] = a9 * b9 * tc;           g.m[3][4] = a9 * b9 * t;             g.m[3][5] = -a9 * ppc;
  const double a10 = (pc2 * s1 - s2 + p2 * s3) * nt_, b10 = ppc * (s1 - s3) * nt_;
  g.m[4][0] = -a10 * t * tc;          g.m[4][1] = -a10 * (t2 - tc2);       g.m[4][2] = a10 * t * tc;
  g.m[4][3] = b10 * t;                g.m[4][4] = -b10 * tc;               g.m[4][5] = 0;
}

IA3_HD void make_geom(const double* x, const FitCfg& cfg, Geom& g) {
  GeomScalars q;
  geom_scalars(x, cfg, q);
  geom_assemble(x, q, g);
}

// signal part f0 = exp(h - xᵀAx/2) at a voxel (coordinates are exact small integers)
IA3_HD double model_f0(const Geom& g, double vz, double vx, double vy) {
  const double xt = vz - g.c[0], yt = vx - g.c[1], zt = vy - g.c[2];
  const double xs = g.q[0] * xt * xt + g.q[1] * yt * yt + g.q[2] * zt * zt + g.q[3] * xt * yt +
                    g.q[4] * xt * zt + g.q[5] * yt * zt;
  return exp(g.h - 0.5 * xs);
}

// Jacobian row (float32-rounded entries, as the reference hands MINPACK a float32 array) and f0
IA3_HD double model_jac(const Geom& g, double vz, double vx, double vy, double* J) {
  const double xt = vz - g.c[0], yt = vx - g.c[1], zt = vy - g.c[2];
  const double mo[6] = {xt * xt, xt * yt, yt * yt, xt * zt, yt * zt, zt * zt};
  const double xs = g.q[0] * mo[0] + g.q[1] * mo[2] + g.q[2] * mo[5] + g.q[3] * mo[1] +
                    g.q[4] * mo[3] + g.q[5] * mo[4];
  const double F = exp(g.h - 0.5 * xs);
  J[0] = (double)(float)g.ebk_j;
  J[1] = (double)(float)F;
#pragma unroll
  for (int k = 0; k < 3; ++k)
    J[2 + k] = (double)(float)(F * (g.l[k][0] * xt + g.l[k][1] * yt + g.l[k][2] * zt));
  // (four coefficients are zero by construction — m[1][3..5]: the second width does not enter the z cross terms,
  // m[4][5]: the in-plane angle does not enter zt² — and their terms, ±0 for finite coordinates, leave every sum as it is)
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    double s = g.m[k][0] * mo[0];
#pragma unroll
    for (int a = 1; a < 6; ++a)
      if (!((k == 1 && a >= 3) || (k == 4 && a == 5))) s += g.m[k][a] * mo[a];
    J[5 + k] = (double)(float)(F * s);
  }
  return F;
}

// natural parameters [h, c0, c1, c2, bk, w0, w1, w2, sin_t, sin_p] (eps appended by the caller)
IA3_HD void to_natural(const double* x, const FitCfg& cfg, float* p) {
  p[0] = (float)exp(x[1]);
  double c[3];
  centers_of(x, cfg, c);
  p[1] = (float)c[0]; p[2] = (float)c[1]; p[3] = (float)c[2];
  p[4] = (float)exp(x[0]);
  p[5] = (float)sqrt(sig_ws(x[5], cfg.min_ws, cfg.max_ws));
  p[6] = (float)sqrt(sig_ws(x[6], cfg.min_ws, cfg.max_ws));
  p[7] = (float)sqrt(sig_ws(x[7], cfg.min_ws, cfg.max_ws));
  p[8] = (float)sig_sine(x[9]);   // t  (from tp)
  p[9] = (float)sig_sine(x[8]);   // p  (from pp)
}

}  // namespace ia3
