// Chained per-FOV spot calling on a resident stack: get_seeds -> firstfit -> repeatfit -> row filters
// (spot_tools/fitting.py:169-237 fit_fov_image without the optional intensity normalisation).
#include "ia3_rt.h"
#include "ia3_pipe.h"
#include <math.h>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <unistd.h>

using namespace ia3rt;

// counters of the calling thread's last ia3_fit_fov_dev: fits run, model evaluations, voxel evaluations
static thread_local long long t_last_stats[5] = {0, 0, 0, 0, 0};

namespace ia3pipe {
int filter_rows(const ia3_stack* im, const float* ps, int n, float* out_rows, int capacity, int* n_rows) {
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const float* r = ps + (size_t)i * 11;
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

// seeds known on the host (count n; centres on the device or on the host)
int fit_known_seeds(const ia3_stack* im, const ia3k::SeedDev& sd, int n, const ia3_fit_params* fp, float* out_rows,
                    int capacity, int* n_rows, int* n_iter, long long* stats5) {
  ia3_fitter* f = nullptr;
  int rc;
  if (sd.on_device) {
    rc = ia3k::fit_create_dev(im, sd.d_zxy, n, fp, &f); if (rc) return rc;   // the seed list never left HBM
  } else {
    std::vector<double> c((size_t)n * 3);
    for (int i = 0; i < n; ++i) { c[3 * i] = sd.host.zxyh[4 * i]; c[3 * i + 1] = sd.host.zxyh[4 * i + 1]; c[3 * i + 2] = sd.host.zxyh[4 * i + 2]; }
    rc = ia3_fit_create(im, c.data(), n, fp, &f); if (rc) return rc;
  }
  return fit_with(f, im, n, out_rows, capacity, n_rows, n_iter, stats5);
}

// firstfit + repeatfit + row filters with a fitter that is ready (destroyed here)
int fit_with(ia3_fitter* f, const ia3_stack* im, int n, float* out_rows, int capacity, int* n_rows, int* n_iter, long long* stats5) {
  int rc;
  static const bool dbg = getenv("IA3_DEBUG_TIMES") != nullptr;
  auto now = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; };
  const double t0 = dbg ? now() : 0;
  static thread_local std::vector<float> ps;   // kept between calls: see ia3_fit_destroy on what a fresh 220 KB vector costs
  ps.resize((size_t)n * 11);
  rc = ia3_fit_run(f);
  const double t1 = dbg ? now() : 0;
  if (!rc) rc = ia3_fit_results_ex(f, ps.data(), nullptr, nullptr, n_iter);
  if (!rc) { ia3k::fit_host_counters(f, t_last_stats); if (stats5) memcpy(stats5, t_last_stats, sizeof(t_last_stats)); }
  const double t2 = dbg ? now() : 0;
  ia3_fit_destroy(f);
  if (rc) return rc;
  rc = filter_rows(im, ps.data(), n, out_rows, capacity, n_rows);
  if (dbg) fprintf(stderr, "fit_known_seeds: create->run done %.1f us, results %.1f us, destroy+filter %.1f us\n", t1 - t0, t2 - t1, now() - t2);
  return rc;
}
}  // namespace ia3pipe
using ia3pipe::filter_rows;

extern "C" int ia3_fit_fov_dev(const ia3_stack* im, const ia3_seed_params* sp, const ia3_fit_params* fp,
                               float* out_rows, int capacity, int* n_rows, int* n_seeds, int* n_iter) {
  int rc = ensure_init(); if (rc) return rc;
  if (!im || !sp || !fp || !n_rows) return set_error(IA3_EINVAL, "null argument");
  if (n_seeds) *n_seeds = 0;
  if (n_iter) *n_iter = 0;
  *n_rows = 0;
  for (int k = 0; k < 5; ++k) t_last_stats[k] = 0;
  // (Leaving the seed COUNT on the device as well — fitter sized for the finish capacity, kernels reading the count —
  // was measured: no gain for one stream, since the host already queues ahead of the device, and 12 % slower with twelve
  // images in flight because every fit launch then carries 16 k mostly empty blocks; profiles/r02b/ab_sync.log.)
  PutDefer defer_puts;   // the seed stage's scratch goes back to the cache behind the first fit launch, not in front of it
  ia3k::SeedDev sd;
  // The fitter is made while the seed count is still on the device: its set-up kernels are queued right behind the seed
  // stage's finish and run while the host waits for the count.  Laid out for 1.5 x the seeds of this thread's previous
  // image (at least 8192); an image with more gets a fitter of the right size afterwards.
  struct Ahead { const ia3_stack* im; const ia3_fit_params* fp; ia3_fitter* f; int cap; int rc; } ah{im, fp, nullptr, 0, 0};
  static thread_local int t_last_n = 0;
  ah.cap = t_last_n + t_last_n / 2 < 8192 ? 8192 : t_last_n + t_last_n / 2;
  sd.ahead_ctx = &ah;
  sd.ahead = [](void* ctx, const double* d_zxy, const unsigned* d_count, int max_num_seeds) {
    Ahead* a = (Ahead*)ctx;
    if (a->f) { ia3_fit_destroy(a->f); a->f = nullptr; }   // the seed stage started over (dense filter after an overflow)
    a->rc = ia3k::fit_create_ahead(a->im, d_zxy, a->cap, d_count, max_num_seeds, a->fp, &a->f);
    if (a->rc) a->f = nullptr;
  };
  dbg_stamp("fit_fov_dev enter");
  rc = ia3k::dog_seed_dev(im, *sp, sd);
  if (rc) { if (ah.f) ia3_fit_destroy(ah.f); return rc; }
  dbg_stamp("seed stage returned");
  const int n = sd.on_device ? sd.n : (int)(sd.host.zxyh.size() / 4);
  if (n_seeds) *n_seeds = n;
  t_last_n = n;
  if (ah.f && (!sd.on_device || n == 0 || n > ah.cap)) { ia3_fit_destroy(ah.f); ah.f = nullptr; }   // not the common case
  if (n == 0) return IA3_OK;  // fitting.py:206-207
  if (!ah.f) return ia3pipe::fit_known_seeds(im, sd, n, fp, out_rows, capacity, n_rows, n_iter, nullptr);
  rc = ia3k::fit_set_count(ah.f, n);
  if (rc) { ia3_fit_destroy(ah.f); return rc; }
  return ia3pipe::fit_with(ah.f, im, n, out_rows, capacity, n_rows, n_iter, nullptr);
}

extern "C" int ia3_fit_fov_wait_share(int64_t* wait_cycles, int64_t* wave_cycles) {
  if (wait_cycles) *wait_cycles = t_last_stats[3];
  if (wave_cycles) *wave_cycles = t_last_stats[4];
  return IA3_OK;
}

extern "C" int ia3_fit_fov_stats(int64_t* fits, int64_t* nfev, int64_t* voxel_evals) {
  if (fits) *fits = t_last_stats[0];
  if (nfev) *nfev = t_last_stats[1];
  if (voxel_evals) *voxel_evals = t_last_stats[2];
  return IA3_OK;
}

// ---- library-owned worker threads ------------------------------------------------------------------------------
// Created on first use and kept for the life of the process (detached, idle on a condition variable): each keeps its
// HIP streams, scratch ordering and pinned staging ring (runtime.cpp: all thread-local), so a batch call costs no
// stream creation or hipHostMalloc.  A forked child starts with an empty pool (threads do not survive fork).
namespace {
struct Pool {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::function<void()>> q;
  int threads = 0;
  pid_t pid = 0;
};
Pool* g_pool = nullptr;        // leaked on purpose: workers may outlive static destruction
std::mutex g_pool_mu;
}  // namespace

void ia3pipe::pool_run(int workers, const std::function<void()>& fn) {
  Pool* p;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (!g_pool || g_pool->pid != getpid()) { g_pool = new Pool(); g_pool->pid = getpid(); }
    p = g_pool;
  }
  // completion state outlives this frame until the last worker has let go of it (shared ownership)
  struct Sync { std::mutex mu; std::condition_variable cv; int left; };
  auto sync = std::make_shared<Sync>();
  sync->left = workers;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    for (; p->threads < workers; ++p->threads)
      std::thread([p] {
        for (;;) {
          std::function<void()> job;
          {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv.wait(lk, [p] { return !p->q.empty(); });
            job = std::move(p->q.front());
            p->q.pop_front();
          }
          job();
        }
      }).detach();
    for (int t = 0; t < workers; ++t)
      p->q.push_back([&fn, sync] {
        fn();   // fn lives in the caller's frame, which waits below until every worker has returned from it
        std::lock_guard<std::mutex> lk(sync->mu);
        if (--sync->left == 0) sync->cv.notify_all();
      });
  }
  p->cv.notify_all();
  std::unique_lock<std::mutex> lk(sync->mu);
  sync->cv.wait(lk, [&] { return sync->left == 0; });
}
using ia3pipe::pool_run;

// A batch of independent FOVs from ONE caller thread.  The reference spreads its per-image tasks over an mp.Pool
// (classes/field_of_view.py:1129-1142).  Here the batch runs as a two-stage pipeline over groups of `in_flight` images:
//   stage A  `in_flight` library threads, each with its own pair of HIP streams (runtime.cpp ThreadCtx) and its own
//            pinned staging ring: upload (host jobs) and get_seeds of one image each; the seed list stays on the device;
//   stage B  one more thread: ONE fitter over all the images of a group (fit.hip, FitArgs: the work list runs over the
//            seeds of every field), while stage A is already seeding the next group.
// What makes a lone image slow — a chain of dependent refits in a crowded territory, a plateau-duplicate seed that refits
// noise to maxfev in every sweep — occupies one wave; in a group fit the other 2047 waves work on the other
// images' seeds meanwhile.  Results are exactly those of ia3_fit_fov_dev job by job (same kernels, same per-seed
// arithmetic; neighbours, Voronoi ties and sweep order never cross an image).
namespace {
struct Slot {
  ia3_stack* up = nullptr;          // uploaded here (host jobs): freed after the group's fit
  const ia3_stack* im = nullptr;
  ia3k::SeedDev sd;
  int n = 0;
  bool seeded = false, fitted = false;
};

}  // namespace

void ia3pipe::fit_group_items(FitItem* items, int n_items, const ia3_fit_params* fp) {
  std::vector<int> members;
  for (int k = 0; k < n_items; ++k)
    if (!items[k].rc && items[k].n > 0) members.push_back(k);
  if (members.empty()) return;
  std::vector<const ia3_stack*> ims;
  std::vector<const double*> seeds;
  std::vector<int> ns;
  for (int k : members) { ims.push_back(items[k].im); seeds.push_back(items[k].d_zxy); ns.push_back(items[k].n); }
  ia3_fitter* f = nullptr;
  int rc = ia3k::fit_create_multi(ims.data(), seeds.data(), ns.data(), (int)members.size(), fp, &f);
  std::vector<float> ps;
  std::vector<int> iters(members.size());
  std::vector<long long> cnt(3 * members.size());
  if (!rc) rc = ia3_fit_run(f);
  if (!rc) {
    size_t tot = 0;
    for (int v : ns) tot += (size_t)v;
    ps.resize(tot * 11);
    rc = ia3_fit_results_ex(f, ps.data(), nullptr, nullptr, nullptr);
  }
  if (!rc) rc = ia3k::fit_fov_results(f, iters.data(), cnt.data());
  if (!rc) {
    const int* st = ia3k::fit_fov_starts(f);
    for (size_t m = 0; m < members.size(); ++m) {
      FitItem& j = items[members[m]];
      j.n_iter = iters[m];
      j.fits = cnt[3 * m]; j.nfev = cnt[3 * m + 1]; j.voxel_evals = cnt[3 * m + 2];
      const int r = filter_rows(ims[m], ps.data() + (size_t)st[m] * 11, ns[m], j.rows, j.capacity, &j.n_rows);
      if (r) { j.rc = r; j.err = ia3_last_error(); }
    }
  } else {
    const std::string msg = ia3_last_error();
    for (int k : members) { items[k].rc = rc; items[k].err = msg; }
  }
  ia3_fit_destroy(f);
}

namespace {
int fit_group(ia3_fov_job* jobs, std::vector<std::unique_ptr<Slot>>& slots, int lo, int hi, const ia3_fit_params* fp,
              std::vector<std::string>& errs) {
  std::vector<ia3pipe::FitItem> items;
  std::vector<int> who;
  for (int k = lo; k < hi; ++k)
    if (!jobs[k].rc && !slots[(size_t)k]->fitted && slots[(size_t)k]->n > 0) {
      ia3pipe::FitItem it;
      it.im = slots[(size_t)k]->im; it.d_zxy = slots[(size_t)k]->sd.d_zxy; it.n = slots[(size_t)k]->n;
      it.rows = jobs[k].rows; it.capacity = jobs[k].capacity;
      items.push_back(it);
      who.push_back(k);
    }
  if (items.empty()) return IA3_OK;
  ia3pipe::fit_group_items(items.data(), (int)items.size(), fp);
  for (size_t m = 0; m < items.size(); ++m) {
    ia3_fov_job& j = jobs[who[m]];
    const ia3pipe::FitItem& it = items[m];
    j.n_rows = it.n_rows; j.n_iter = it.n_iter;
    j.fits = it.fits; j.nfev = it.nfev; j.voxel_evals = it.voxel_evals;
    if (it.rc) { j.rc = it.rc; errs[(size_t)who[m]] = it.err; }
  }
  return IA3_OK;
}
}  // namespace

extern "C" int ia3_fit_fovs(ia3_fov_job* jobs, int n_jobs, int dtype, int Z, int X, int Y, const ia3_seed_params* sp,
                            const ia3_fit_params* fp, int in_flight) {
  int rc = ensure_init(); if (rc) return rc;
  if (n_jobs < 0 || (n_jobs > 0 && !jobs) || !sp || !fp) return set_error(IA3_EINVAL, "null argument");
  for (int k = 0; k < n_jobs; ++k)
    if (!jobs[k].host && !jobs[k].dev) return set_error(IA3_EINVAL, "job %d has neither a host nor a resident stack", k);
  if (n_jobs == 0) return IA3_OK;
  if (in_flight <= 0) in_flight = 4;
  if (in_flight > 64) in_flight = 64;
  if (in_flight > n_jobs) in_flight = n_jobs;
  std::vector<std::string> errs((size_t)n_jobs);
  for (int k = 0; k < n_jobs; ++k) {
    ia3_fov_job& j = jobs[k];
    j.n_rows = j.n_seeds = j.n_iter = 0;
    j.fits = j.nfev = j.voxel_evals = 0;
    j.rc = 0;
  }
  if (in_flight <= 1) {   // one image at a time: the plain per-FOV path
    for (int k = 0; k < n_jobs; ++k) {
      ia3_fov_job& j = jobs[k];
      ia3_stack* up = nullptr;
      int r = IA3_OK;
      if (!j.dev) r = ia3_stack_upload(j.host, dtype, Z, X, Y, &up);
      if (!r) r = ia3_fit_fov_dev(j.dev ? j.dev : up, sp, fp, j.rows, j.capacity, &j.n_rows, &j.n_seeds, &j.n_iter);
      if (!r) { j.fits = t_last_stats[0]; j.nfev = t_last_stats[1]; j.voxel_evals = t_last_stats[2]; }
      if (up) ia3_stack_free(up);
      j.rc = r;
      if (r) errs[(size_t)k] = ia3_last_error();
    }
  } else {
    // Resident stacks may still be in production on the CALLER's stream (a correction chain queued just before this
    // call); the seeding threads have streams of their own, so each of them first waits for what the caller has queued
    // up to here.  (Without this a caller had to ia3_sync() first; one that did not seeded half-warped images as soon
    // as another host thread kept the device busy.)
    hipEvent_t caller_done = nullptr;
    bool any_resident = false;
    for (int k = 0; k < n_jobs; ++k) any_resident = any_resident || jobs[k].dev;
    if (any_resident) {
      IA3_HIP(hipEventCreateWithFlags(&caller_done, hipEventDisableTiming));
      if (hipEventRecord(caller_done, stream()) != hipSuccess) {
        (void)hipEventDestroy(caller_done);
        return set_error(IA3_EHIP, "event record on the caller's stream failed");
      }
    }
    const int G = in_flight < ia3k::fit_max_fovs() ? in_flight : ia3k::fit_max_fovs();   // images per group fit
    const int W = in_flight < 16 ? in_flight : 16;                                       // seeding threads
    const int groups = (n_jobs + G - 1) / G;
    std::vector<std::unique_ptr<Slot>> slots((size_t)n_jobs);
    for (auto& s : slots) s.reset(new Slot());
    std::mutex mu;
    std::condition_variable cv;
    int groups_fitted = 0;                 // stage A runs at most AHEAD groups ahead of stage B (resident stacks)
    std::atomic<int> next{0};
    // TWO fitter threads take the groups in turn: what ends a group fit is a handful of waves — plateau twins that refit
    // noise to maxfev, the last links of a dependency chain — while the other 2 000 have run out of tickets; the next
    // group's fit runs on the SIMDs they left (uint16 FOVs in groups of 32: 2.2 -> 1.9 ms each)
    const int F = groups > 1 ? 2 : 1;
    const int AHEAD = F + 1;
    std::atomic<int> fitters{0};
    auto stage_a = [&]() {
      int init_rc = ensure_init();   // this thread's streams
      if (!init_rc && caller_done && hipStreamWaitEvent(stream(), caller_done, 0) != hipSuccess)
        init_rc = set_error(IA3_EHIP, "wait for the caller's stream failed");
      for (;;) {
        const int k = next.fetch_add(1);
        if (k >= n_jobs) break;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return k / G < groups_fitted + AHEAD; });
        }
        ia3_fov_job& j = jobs[k];
        Slot& s = *slots[(size_t)k];
        int r = init_rc;
        if (!r && !j.dev) r = ia3_stack_upload(j.host, dtype, Z, X, Y, &s.up);
        s.im = j.dev ? j.dev : s.up;
        if (!r) r = ia3k::dog_seed_dev(s.im, *sp, s.sd);
        if (!r) {
          s.n = s.sd.on_device ? s.sd.n : (int)(s.sd.host.zxyh.size() / 4);
          j.n_seeds = s.n;
          if (s.n > 0 && !s.sd.on_device) {   // the rare host-side seed finish (> 32768 candidates): fitted here, on its own
            r = ia3pipe::fit_known_seeds(s.im, s.sd, s.n, fp, j.rows, j.capacity, &j.n_rows, &j.n_iter, nullptr);
            if (!r) { j.fits = t_last_stats[0]; j.nfev = t_last_stats[1]; j.voxel_evals = t_last_stats[2]; }
            s.fitted = true;
          }
        }
        if (r) { j.rc = r; errs[(size_t)k] = ia3_last_error(); }
        {
          std::lock_guard<std::mutex> lk(mu);
          s.seeded = true;
        }
        cv.notify_all();
      }
    };
    auto stage_b = [&](int first) {
      (void)ensure_init();
      for (int g = first; g < groups; g += F) {
        const int lo = g * G, hi = lo + G < n_jobs ? lo + G : n_jobs;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { for (int k = lo; k < hi; ++k) if (!slots[(size_t)k]->seeded) return false; return true; });
        }
        fit_group(jobs, slots, lo, hi, fp, errs);
        for (int k = lo; k < hi; ++k) {   // the group's stacks and seed lists go back to the scratch cache
          Slot& s = *slots[(size_t)k];
          if (s.up) { ia3_stack_free(s.up); s.up = nullptr; }
          slots[(size_t)k].reset(new Slot());
          slots[(size_t)k]->seeded = true;
        }
        {
          std::lock_guard<std::mutex> lk(mu);
          ++groups_fitted;
        }
        cv.notify_all();
      }
    };
    pool_run(W + F, [&]() { const int me = fitters.fetch_add(1); if (me < F) stage_b(me); else stage_a(); });
    if (caller_done) (void)hipEventDestroy(caller_done);
  }
  for (int k = 0; k < n_jobs; ++k)
    if (jobs[k].rc) return set_error(jobs[k].rc, "FOV %d: %s", k, errs[(size_t)k].c_str());
  return IA3_OK;
}
