// Chained per-FOV spot calling on a resident stack: get_seeds -> firstfit -> repeatfit -> row filters
// (spot_tools/fitting.py:169-237 fit_fov_image without the optional intensity normalisation).
#include "ia3_rt.h"
#include <math.h>
#include <string.h>

using namespace ia3rt;

extern "C" int ia3_fit_fov_dev(const ia3_stack* im, const ia3_seed_params* sp, const ia3_fit_params* fp,
                               float* out_rows, int capacity, int* n_rows, int* n_seeds, int* n_iter) {
  int rc = ensure_init(); if (rc) return rc;
  if (!im || !sp || !fp || !n_rows) return set_error(IA3_EINVAL, "null argument");
  ia3k::SeedDev sd;
  rc = ia3k::dog_seed_dev(im, *sp, sd); if (rc) return rc;
  const int n = sd.on_device ? sd.n : (int)(sd.host.zxyh.size() / 4);
  if (n_seeds) *n_seeds = n;
  if (n_iter) *n_iter = 0;
  *n_rows = 0;
  if (n == 0) return IA3_OK;  // fitting.py:206-207
  ia3_fitter* f = nullptr;
  if (sd.on_device) {
    rc = ia3k::fit_create_dev(im, sd.d_zxy, n, fp, &f); if (rc) return rc;   // the seed list never left HBM
  } else {
    std::vector<double> c((size_t)n * 3);
    for (int i = 0; i < n; ++i) { c[3 * i] = sd.host.zxyh[4 * i]; c[3 * i + 1] = sd.host.zxyh[4 * i + 1]; c[3 * i + 2] = sd.host.zxyh[4 * i + 2]; }
    rc = ia3_fit_create(im, c.data(), n, fp, &f); if (rc) return rc;
  }
  std::vector<float> ps((size_t)n * 11);
  rc = ia3_fit_run(f);
  if (!rc) rc = ia3_fit_results_ex(f, ps.data(), nullptr, nullptr, n_iter);
  ia3_fit_destroy(f);
  if (rc) return rc;
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const float* r = &ps[(size_t)i * 11];
    bool ok = true;
    for (int k = 0; k < 11; ++k) if (isnan(r[k])) ok = false;                       // :232
    if (ok) ok = r[1] > 0 && r[2] > 0 && r[3] > 0 && r[1] < im->Z && r[2] < im->X && r[3] < im->Y;  // :235-236
    if (!ok) continue;
    if (m < capacity && out_rows) memcpy(out_rows + (size_t)m * 11, r, 11 * sizeof(float));
    ++m;
  }
  *n_rows = m;
  if (m > capacity) return set_error(IA3_ECAPACITY, "row buffer too small: need %d rows", m);
  return IA3_OK;
}
