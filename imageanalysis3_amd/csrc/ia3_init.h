// Start point of GaussianFit (External/Fitting_v4.py:175-185): log of the mean of the 10 smallest
// / 10 largest voxel values, offsets 0, widths init_w, angles 0 — every entry rounded to float32
// (the reference builds p_ as a float32 array).  The means reproduce NumPy's summation order
// for a contiguous 10-vector (8-lane pairwise block + 2 tail adds) in the dtype NumPy would use:
//   kind 0: float32 data  -> float32 accumulation, float32 division
//   kind 1: integer data  -> exact float64
//   kind 2: float64 data  -> float64 pairwise
#pragma once
#include "ia3_model.h"

namespace ia3 {

template <class T>
IA3_HD T np_sum10(const T* a) {
  T r = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  r = r + a[8];
  r = r + a[9];
  return r;
}

// lo10 / hi10: the ten smallest / largest values, each ascending.
IA3_HD void init_guess(const double* lo10, const double* hi10, int kind, const FitCfg& cfg, double* x) {
  double mlo, mhi;
  if (kind == 0) {
    float a[10], b[10];
    for (int k = 0; k < 10; ++k) { a[k] = (float)lo10[k]; b[k] = (float)hi10[k]; }
    mlo = (double)(np_sum10<float>(a) / 10.0f);
    mhi = (double)(np_sum10<float>(b) / 10.0f);
  } else {
    mlo = np_sum10<double>(lo10) / 10.0;
    mhi = np_sum10<double>(hi10) / 10.0;
  }
  const double eps = 4.5399929762484854e-05;  // np.exp(-10.)
  double bk = log(mlo > eps ? mlo : eps);
  double h = log(mhi > eps ? mhi : eps);
  double wsq = cfg.init_w * cfg.init_w;
  double wg = log((cfg.max_ws - wsq) / (wsq - cfg.min_ws));
  x[0] = (double)(float)bk; x[1] = (double)(float)h;
  x[2] = 0; x[3] = 0; x[4] = 0;
  x[5] = x[6] = x[7] = (double)(float)wg;
  if (cfg.variant == 1) { x[5] = (double)(float)cfg.iw[0]; x[6] = (double)(float)cfg.iw[1]; x[7] = (double)(float)cfg.iw[2]; }
  x[8] = 0; x[9] = 0;
}

}  // namespace ia3
