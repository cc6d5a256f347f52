// Internal: pieces of the batch pipelines (pipeline.cpp) shared by ia3_fit_fovs and ia3_process_movies (movie.cpp).
#pragma once
#include "ia3_rt.h"
#include <functional>
struct ia3_fitter;
#include <string>

namespace ia3pipe {

// spot_tools/fitting.py:232-237: drop NaN rows and centres outside the image; *n_rows = rows that pass (IA3_ECAPACITY when
// they do not fit into `capacity`)
int filter_rows(const ia3_stack* im, const float* ps, int n, float* out_rows, int capacity, int* n_rows);

// firstfit + repeatfit + row filters of ONE image whose seeds are known (on the device or on the host); counters of the
// fit in stats5 (fits, evaluations, voxel evaluations, wait cycles, wave cycles) when not NULL
int fit_known_seeds(const ia3_stack* im, const ia3k::SeedDev& sd, int n, const ia3_fit_params* fp, float* out_rows,
                    int capacity, int* n_rows, int* n_iter, long long* stats5);

int fit_with(ia3_fitter* f, const ia3_stack* im, int n, float* out_rows, int capacity, int* n_rows, int* n_iter, long long* stats5);

// One image of a group fit: its resident stack, its n seeds (n x 3 float64 on the device) and where its table goes.
struct FitItem {
  const ia3_stack* im = nullptr;
  const double* d_zxy = nullptr;
  int n = 0;
  float* rows = nullptr;
  int capacity = 0;
  int n_rows = 0, n_iter = 0, rc = 0;          // out
  long long fits = 0, nfev = 0, voxel_evals = 0;   // out
  std::string err;                              // out: message when rc != 0
};
// ONE fitter over the seeds of all items (same shape and dtype, at most ia3k::fit_max_fovs(); items with n == 0 are
// skipped): neighbours, Voronoi ties and sweep order never cross an image, tables are those of one fit per image.
// Must run on a thread whose stream is ordered after whatever produced the stacks and the seed lists.
void fit_group_items(FitItem* items, int n_items, const ia3_fit_params* fp);

// fn on `workers` library-owned threads at once (created on first use, kept for the life of the process, each with its
// own HIP streams, scratch ordering and pinned staging ring); returns when all of them have returned
void pool_run(int workers, const std::function<void()>& fn);

}  // namespace ia3pipe
