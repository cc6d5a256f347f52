// Whole round-folder movies through one pipelined call (ia3_process_movies) and align_image's crop loop + consensus on
// resident stacks (ia3_align_image_dev).
//
// Reference: classes/batch_functions.py:60-302 (batch_process_image_to_spots: correct_fov_image -> fit_fov_image per
// channel), fanned out per movie over an mp.Pool by classes/field_of_view.py:1027-1142; the chain itself is
// io_tools/load.py:166-522, the drift correction_tools/alignment.py:527-695.
//
// Why a pipeline: one movie is 1.68 GB of uint16 from the host (30 ms of PCIe), ~35 ms of corrections / drift / cubic
// warps on the device and three uint16 images to fit, and a uint16 image fitted on its own waits tens of milliseconds
// for a handful of fits that run to maxfev.  Serially per host thread that is 0.12 s per movie.  Here
//   U   one thread uploads movie k+1 (at most `upload_ahead` raw movies wait in HBM),
//   C   `correct_threads` threads, each on its own streams, take the uploaded movies in order: split channels, hot
//       pixels, z shift, bleedthrough, illumination, bead drift (host round trips for the correlation peaks: with two
//       or three movies in this stage the device always has another movie's warps to run meanwhile), cubic warps,
//       high-pass, get_seeds of every selected channel, download of the corrected images if the caller wants them,
//   F   one thread fits the seeded images of SEVERAL movies with one group fitter (fit.hip: the straggling fits of one
//       image run beside the other images' fits).
// Every kernel and every per-image operation is the one the per-movie path runs: tables, drifts and images are identical.
#include "ia3_rt.h"
#include "ia3_pipe.h"
#include <math.h>
#include <string.h>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

using namespace ia3rt;

namespace {

double now_ms() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

// np.nanmean(d[idx], axis=0): NumPy adds the rows one after the other
void nanmean_rows(const double* d, const int* idx, int n, double out[3]) {
  for (int a = 0; a < 3; ++a) {
    double s = 0;
    int c = 0;
    for (int k = 0; k < n; ++k) {
      const double v = d[3 * idx[k] + a];
      if (!isnan(v)) { s += v; ++c; }
    }
    out[a] = c ? s / (double)c : NAN;
  }
}
double dist3(const double* a, const double* b) {   // np.linalg.norm / scipy pdist: sqrt(((dx^2 + dy^2) + dz^2))
  double s = 0;
  for (int k = 0; k < 3; ++k) { const double t = a[k] - b[k]; s += t * t; }
  return sqrt(s);
}

// correction_tools/alignment.py:664-674: with >= min_good drifts in, those within diff_th of the mean of all of them;
// if there are >= min_good of those, their mean is the answer
bool consensus(const double* d, int n, int min_good, double diff_th, double out[3]) {
  if (n < min_good) return false;
  std::vector<int> all((size_t)n), close;
  for (int k = 0; k < n; ++k) all[(size_t)k] = k;
  double centre[3];
  nanmean_rows(d, all.data(), n, centre);
  for (int k = 0; k < n; ++k)
    if (dist3(d + 3 * k, centre) <= diff_th) close.push_back(k);
  if ((int)close.size() < min_good) return false;
  nanmean_rows(d, close.data(), (int)close.size(), out);
  return true;
}
// :676-693: no agreement — the two drifts closest to each other and the one nearest to both of them
void closest_three(const double* d, int n, double out[3]) {
  if (n == 1) { memcpy(out, d, 3 * sizeof(double)); return; }
  std::vector<double> g((size_t)n * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) g[(size_t)i * n + j] = i == j ? INFINITY : dist3(d + 3 * i, d + 3 * j);
  int bi = 0, bj = 0, third = 0;
  for (int i = 0; i < n; ++i)           // np.argmin over the flattened matrix: first smallest entry in row-major order
    for (int j = 0; j < n; ++j)
      if (g[(size_t)i * n + j] < g[(size_t)bi * n + bj]) { bi = i; bj = j; }
  auto both = [&](int k) { return g[(size_t)k * n + bi] + g[(size_t)k * n + bj]; };
  for (int k = 1; k < n; ++k)
    if (both(k) < both(third)) third = k;
  const int idx[3] = {bi, bj, third};
  nanmean_rows(d, idx, 3, out);
}

}  // namespace

// align_image's loop over the crops of a DriftRef: the first min_good crops (at most three) in one batch — the rule
// cannot stop before it has that many — then crop by crop until enough of them agree
static int align_with_ref(const ia3_stack* src, ia3k::DriftRef* dr, int n_crops, int upsample, int normalization, int min_good_drifts,
                          double drift_diff_th, double* drift, int* flag, double* drifts_out, int* n_used) {
  std::vector<double> d;
  bool agreed = false;
  int done = 0;
  while (done < n_crops && !agreed) {
    int count = done == 0 ? (min_good_drifts < 3 ? (min_good_drifts < 1 ? 1 : min_good_drifts) : 3) : 1;
    if (count > n_crops - done) count = n_crops - done;
    double sh[9];
    int rc = ia3k::drift_crops(src, dr, done, count, upsample, normalization, sh); if (rc) return rc;
    for (int k = 0; k < count && !agreed; ++k) {   // the reference looks at the rule after every crop
      d.insert(d.end(), sh + 3 * k, sh + 3 * k + 3);
      agreed = consensus(d.data(), (int)(d.size() / 3), min_good_drifts, drift_diff_th, drift);
    }
    done += count;
  }
  const int n = (int)(d.size() / 3);
  if (!agreed) closest_three(d.data(), n, drift);
  if (flag) *flag = agreed ? 0 : 1;
  if (drifts_out) memcpy(drifts_out, d.data(), d.size() * sizeof(double));
  if (n_used) *n_used = n;
  return IA3_OK;
}

extern "C" int ia3_align_image_dev(const ia3_stack* src, const ia3_stack* ref, const int* crops, int n_crops, int upsample,
                                   int normalization, int min_good_drifts, double drift_diff_th, double* drift, int* flag,
                                   double* drifts_out, int* n_used) {
  int rc = ensure_init(); if (rc) return rc;
  if (!src || !ref || !crops || !drift || n_crops < 1) return set_error(IA3_EINVAL, "null argument");
  if (src->Z != ref->Z || src->X != ref->X || src->Y != ref->Y)
    return set_error(IA3_EINVAL, "shape of target image and reference image doesnt match");
  if (src->dtype != ref->dtype) return set_error(IA3_EINVAL, "source and reference stacks differ in dtype");
  if (n_crops > 8) return set_error(IA3_EINVAL, "at most 8 drift crops");
  ia3k::DriftRef* dr = nullptr;
  rc = ia3k::drift_ref_create(ref, crops, n_crops, false, &dr); if (rc) return rc;   // spectra of the crops that get used
  rc = align_with_ref(src, dr, n_crops, upsample, normalization, min_good_drifts, drift_diff_th, drift, flag, drifts_out, n_used);
  ia3k::drift_ref_free(dr);
  return rc;
}

struct ia3_drift_ref { ia3k::DriftRef* r; int n_crops; };

extern "C" int ia3_drift_ref_create(const ia3_stack* ref, const int* crops, int n_crops, ia3_drift_ref** out) {
  if (!out) return set_error(IA3_EINVAL, "null argument");
  ia3k::DriftRef* r = nullptr;
  int rc = ia3k::drift_ref_create(ref, crops, n_crops, true, &r); if (rc) return rc;
  *out = new ia3_drift_ref{r, n_crops};
  return IA3_OK;
}
extern "C" void ia3_drift_ref_free(ia3_drift_ref* r) {
  if (!r) return;
  ia3k::drift_ref_free(r->r);
  delete r;
}
extern "C" int ia3_align_image_ref(const ia3_stack* src, ia3_drift_ref* ref, int upsample, int normalization, int min_good_drifts,
                                   double drift_diff_th, double* drift, int* flag, double* drifts_out, int* n_used) {
  int rc = ensure_init(); if (rc) return rc;
  if (!src || !ref || !drift) return set_error(IA3_EINVAL, "null argument");
  return align_with_ref(src, ref->r, ref->n_crops, upsample, normalization, min_good_drifts, drift_diff_th, drift, flag, drifts_out, n_used);
}

// ---- the movie pipeline -------------------------------------------------------------------------------------------
namespace {

struct Image {                       // one selected channel of one movie, from the corrections to the fit
  int movie = 0, sel = 0;
  ia3_stack* st = nullptr;
  ia3k::SeedDev sd;
  int n = 0;
  bool host_fitted = false;          // seeds came back on the host (> 32768 candidates): fitted by the corrector itself
  std::atomic<int> users{0};         // the fit and the optional download; the last one frees the stack
};

void release(Image* im) {
  if (im->users.fetch_sub(1) == 1) {
    if (im->st) ia3_stack_free(im->st);
    im->st = nullptr;
    // the seed list's scratch block (SeedDev::hold) goes back with the Image itself
    delete im;
  }
}

struct Pipe {
  ia3_movie_job* jobs;
  int n_jobs;
  const ia3_movie_params* p;
  double t0;
  ia3_drift_ref* dref = nullptr;     // reference crop spectra (the caller's, or made for this call)
  std::mutex mu;
  std::condition_variable cv;
  std::vector<ia3_stack*> raw;       // uploaded movies waiting for a corrector
  std::vector<char> uploaded;        // 0 waiting, 1 ready, 2 failed
  int taken = 0;                     // movies a corrector has started on
  int corrected = 0;                 // movies whose images have all reached the fit queue (or failed)
  std::deque<Image*> fitq;
  std::vector<std::string> errs;
  std::atomic<int> next_correct{0};
  std::atomic<int> role{0};
  int group = 12;                    // images per group fit
};

void fail(Pipe& P, int k, int rc) {
  std::lock_guard<std::mutex> lk(P.mu);
  if (!P.jobs[k].rc) { P.jobs[k].rc = rc; P.errs[(size_t)k] = ia3_last_error(); }
}

void uploader(Pipe& P) {
  const ia3_movie_params& p = *P.p;
  int init_rc = ensure_init();
  const int ahead = p.upload_ahead > 0 ? p.upload_ahead : 2;
  for (int k = 0; k < P.n_jobs; ++k) {
    {
      std::unique_lock<std::mutex> lk(P.mu);
      P.cv.wait(lk, [&] { return k - P.taken < ahead; });
    }
    ia3_movie_job& j = P.jobs[k];
    const double ta = now_ms();
    ia3_stack* st = nullptr;
    int rc = init_rc;
    if (!rc) {
      if (j.host_raw) rc = ia3_stack_upload(j.host_raw, IA3_U16, p.frames, p.X, p.Y, &st);
      else if (j.path) rc = ia3_stack_load_file(j.path, j.offset_bytes, p.frames, p.X, p.Y, j.big_endian, &st);
      else rc = set_error(IA3_EINVAL, "movie %d has neither a host array nor a file", k);
    }
    j.t_upload_ms = now_ms() - ta;
    j.stamps[0] = ta - P.t0; j.stamps[1] = now_ms() - P.t0;
    if (rc) fail(P, k, rc);
    {
      std::lock_guard<std::mutex> lk(P.mu);
      P.raw[(size_t)k] = st;
      P.uploaded[(size_t)k] = rc ? 2 : 1;
    }
    P.cv.notify_all();
  }
}

// io_tools/load.py:303-498 on resident stacks: raw movie -> the selected, corrected channels (sel_out), drift and flag
int correct_movie(const ia3_movie_params& p, ia3_drift_ref* dref, ia3_movie_job& j, ia3_stack* raw, ia3_stack** sel_out) {
  std::vector<ia3_stack*> ch((size_t)p.n_load, nullptr);
  std::vector<ia3_stack*> extra;   // replaced stacks (inputs of the bleedthrough mix, unwarped images)
  int rc = IA3_OK;
  auto cleanup = [&](bool keep_sel) {
    for (size_t c = 0; c < ch.size(); ++c) {
      bool is_sel = false;
      if (keep_sel) for (int s = 0; s < p.n_sel; ++s) is_sel = is_sel || p.sel[s] == (int)c;
      if (ch[c] && !is_sel) ia3_stack_free(ch[c]);
    }
    for (ia3_stack* s : extra) ia3_stack_free(s);
  };
  for (int c = 0; c < p.n_load && !rc; ++c)                                   // :303-320 split_im_by_channels
    rc = ia3_stack_deinterleave(raw, p.load_start[c], p.load_step, p.Z, &ch[(size_t)c]);
  if (rc) { cleanup(false); return rc; }
  if (p.hot_pixel_corr)                                                        // :323-334
    for (int c = 0; c < p.n_load && !rc; ++c) rc = ia3_remove_hot_pixels_dev(ch[(size_t)c], 0.5, p.hot_pixel_th, 1, nullptr);
  if (p.z_shift_corr)                                                          // :337-345
    for (int c = 0; c < p.n_load && !rc; ++c) rc = ia3_z_shift_correction_dev(ch[(size_t)c], ch[(size_t)c]);
  if (!rc && p.n_bleed > 0) {                                                  // :348-370
    std::vector<ia3_stack*> ins((size_t)p.n_bleed), outs((size_t)p.n_bleed, nullptr);
    for (int b = 0; b < p.n_bleed && !rc; ++b) {
      ins[(size_t)b] = ch[(size_t)p.bleed_idx[b]];
      rc = ia3_stack_alloc(IA3_U16, p.Z, p.X, p.Y, &outs[(size_t)b]);
    }
    if (!rc) rc = ia3_bleedthrough_correct_dev(ins.data(), p.n_bleed, p.bleed_profile, p.bleed_dtype, outs.data());
    for (int b = 0; b < p.n_bleed; ++b) {
      if (!outs[(size_t)b]) continue;
      if (!rc) { extra.push_back(ch[(size_t)p.bleed_idx[b]]); ch[(size_t)p.bleed_idx[b]] = outs[(size_t)b]; }
      else ia3_stack_free(outs[(size_t)b]);
    }
  }
  for (int c = 0; c < p.n_load && !rc; ++c)                                    // :373-384
    if (p.illum_profile[c]) rc = ia3_illumination_correct_dev(ch[(size_t)c], p.illum_profile[c], p.illum_dtype[c], ch[(size_t)c]);
  // :387-417 drift
  double drift[3] = {j.drift_in[0], j.drift_in[1], j.drift_in[2]};
  int flag = 0;
  if (!rc && j.measure_drift && p.drift_idx >= 0) {
    if (!dref) rc = set_error(IA3_EINVAL, "no reference bead stack");
    else rc = ia3_align_image_ref(ch[(size_t)p.drift_idx], dref, p.precision_fold, p.normalization, p.min_good_drifts, p.drift_diff_th,
                                  drift, &flag, nullptr, nullptr);
  }
  if (rc) { cleanup(false); return rc; }
  memcpy(j.drift, drift, sizeof(drift));
  j.drift_flag = flag;
  const bool any_drift = drift[0] != 0 || drift[1] != 0 || drift[2] != 0;
  if (p.warp)                                                                  // :424-453
    for (int s = 0; s < p.n_sel && !rc; ++s) {
      const int c = p.sel[s];
      if (!(p.warp_always[s] || any_drift)) continue;
      ia3_stack* out = nullptr;
      rc = ia3_stack_alloc(IA3_U16, p.Z, p.X, p.Y, &out);
      if (!rc) rc = ia3_warp3d_dev(ch[(size_t)c], drift, p.chrom_field[s], p.chrom_field[s] ? p.chrom_dtype[s] : 0, 3, IA3_MODE_NEAREST, 0.0, out);
      if (rc) { if (out) ia3_stack_free(out); break; }
      extra.push_back(ch[(size_t)c]);
      ch[(size_t)c] = out;
    }
  if (!rc && p.highpass_sigma > 0)                                             // :489-498 (the selected channels are what leaves)
    for (int s = 0; s < p.n_sel && !rc; ++s) {
      const int c = p.sel[s];
      ia3_stack* out = nullptr;
      rc = ia3_stack_alloc(IA3_U16, p.Z, p.X, p.Y, &out);
      if (!rc) rc = ia3_gaussian_highpass_dev(ch[(size_t)c], p.highpass_sigma, p.highpass_truncate, nullptr, 0, out);
      if (rc) { if (out) ia3_stack_free(out); break; }
      extra.push_back(ch[(size_t)c]);
      ch[(size_t)c] = out;
    }
  if (rc) { cleanup(false); return rc; }
  for (int s = 0; s < p.n_sel; ++s) sel_out[s] = ch[(size_t)p.sel[s]];
  cleanup(true);
  return IA3_OK;
}

// spot_tools/fitting.py:240-258: heights over the image's background level, or over each spot's local one
int normalize_rows(const ia3_movie_params& p, const ia3_stack* im, float* rows, int n_rows) {
  if (!p.normalize || n_rows == 0) return IA3_OK;
  if (p.normalize == 1) {
    double back = 0;
    int r = ia3_find_background_dev(im, p.bg_edges, p.bg_n_edges, p.bg_max_iter, &back); if (r) return r;
    for (int q = 0; q < n_rows; ++q) rows[(size_t)q * 11] = (float)((double)rows[(size_t)q * 11] / back);
    return IA3_OK;
  }
  std::vector<float> cen((size_t)n_rows * 3);
  std::vector<double> backs((size_t)n_rows);
  for (int q = 0; q < n_rows; ++q) memcpy(&cen[(size_t)q * 3], rows + (size_t)q * 11 + 1, 3 * sizeof(float));
  int r = ia3_local_background_dev(im, cen.data(), n_rows, p.bg_crop_size, p.bg_edges, p.bg_n_edges, p.bg_max_iter, backs.data());
  if (r) return r;
  for (int q = 0; q < n_rows; ++q) rows[(size_t)q * 11] = (float)((double)rows[(size_t)q * 11] / backs[(size_t)q]);
  return IA3_OK;
}

void corrector(Pipe& P) {
  const ia3_movie_params& p = *P.p;
  int init_rc = ensure_init();
  for (;;) {
    const int k = P.next_correct.fetch_add(1);
    if (k >= P.n_jobs) break;
    ia3_stack* raw = nullptr;
    bool ok;
    {
      std::unique_lock<std::mutex> lk(P.mu);
      // corrected images wait in HBM for the fitter (0.42 GB each): no more than three groups ahead of it
      P.cv.wait(lk, [&] { return P.uploaded[(size_t)k] != 0 && (int)P.fitq.size() <= 3 * P.group; });
      ok = P.uploaded[(size_t)k] == 1;
      raw = P.raw[(size_t)k];
      P.raw[(size_t)k] = nullptr;
      ++P.taken;
    }
    P.cv.notify_all();
    ia3_movie_job& j = P.jobs[k];
    const double ta = now_ms();
    ia3_stack* sel[IA3_MOVIE_MAXCH] = {};
    int rc = ok ? init_rc : IA3_EINVAL;
    if (ok && !rc) rc = correct_movie(p, P.dref, j, raw, sel);
    if (raw) ia3_stack_free(raw);
    if (ok && rc) fail(P, k, rc);
    std::vector<Image*> ims;
    if (ok && !rc) {
      const bool want_fit = p.fit_spots != 0;
      for (int s = 0; s < p.n_sel; ++s) {
        Image* im = new Image();
        im->movie = k; im->sel = s; im->st = sel[s];
        im->users.store((want_fit ? 1 : 0) + (j.images_out[s] ? 1 : 0) + 1);   // + this loop's own hold
        ims.push_back(im);
      }
      if (want_fit)
        for (Image* im : ims) {                                              // get_seeds of every selected channel
          int r = ia3k::dog_seed_dev(im->st, p.seed[im->sel], im->sd);
          if (!r) {
            im->n = im->sd.on_device ? im->sd.n : (int)(im->sd.host.zxyh.size() / 4);
            j.n_seeds[im->sel] = im->n;
            if (im->n > 0 && !im->sd.on_device) {   // the rare host-side seed finish: fitted here, on its own
              r = ia3pipe::fit_known_seeds(im->st, im->sd, im->n, &p.fit, j.rows[im->sel], j.capacity[im->sel], &j.n_rows[im->sel],
                                           &j.n_iter[im->sel], nullptr);
              if (!r) r = normalize_rows(p, im->st, j.rows[im->sel], j.n_rows[im->sel]);
              im->host_fitted = true;
            }
          }
          if (r) { fail(P, k, r); im->n = 0; im->host_fitted = true; }
        }
      // this thread's stream has produced the stacks and the seed lists; the fitter runs on another stream
      if (want_fit) { int r = stream_wait_spin(stream()); if (r) fail(P, k, r); }
      j.t_correct_ms = now_ms() - ta;
      j.stamps[2] = ta - P.t0; j.stamps[3] = now_ms() - P.t0;
      if (want_fit) {
        std::lock_guard<std::mutex> lk(P.mu);
        for (Image* im : ims) P.fitq.push_back(im);
      }
    }
    {
      std::lock_guard<std::mutex> lk(P.mu);
      ++P.corrected;
    }
    P.cv.notify_all();
    for (Image* im : ims) {                                                  // corrected images back to the host, if wanted
      if (j.images_out[im->sel]) {
        const int r = ia3_stack_download(im->st, j.images_out[im->sel]);
        if (r) fail(P, k, r);
        release(im);
      }
      release(im);   // the loop's own hold
    }
  }
}

void fitter(Pipe& P) {
  const ia3_movie_params& p = *P.p;
  (void)ensure_init();
  const int G = P.group;
  for (;;) {
    std::vector<Image*> grp;
    {
      std::unique_lock<std::mutex> lk(P.mu);
      P.cv.wait(lk, [&] { return (int)P.fitq.size() >= G || P.corrected == P.n_jobs; });
      while (!P.fitq.empty() && (int)grp.size() < G) { grp.push_back(P.fitq.front()); P.fitq.pop_front(); }
      if (grp.empty() && P.corrected == P.n_jobs) return;
    }
    P.cv.notify_all();   // correctors held back by a long queue
    const double ta = now_ms();
    std::vector<ia3pipe::FitItem> items;
    std::vector<Image*> who;
    for (Image* im : grp) {
      if (im->host_fitted || im->n == 0) continue;
      ia3_movie_job& j = P.jobs[im->movie];
      ia3pipe::FitItem it;
      it.im = im->st; it.d_zxy = im->sd.d_zxy; it.n = im->n;
      it.rows = j.rows[im->sel]; it.capacity = j.capacity[im->sel];
      items.push_back(it);
      who.push_back(im);
    }
    if (!items.empty()) ia3pipe::fit_group_items(items.data(), (int)items.size(), &p.fit);
    if (p.normalize)
      for (size_t m = 0; m < items.size(); ++m) {
        ia3pipe::FitItem& it = items[m];
        if (it.rc || it.n_rows == 0) continue;
        const int r = normalize_rows(p, it.im, it.rows, it.n_rows);
        if (r) { it.rc = r; it.err = ia3_last_error(); }
      }
    const double dt = now_ms() - ta;
    for (size_t m = 0; m < items.size(); ++m) {
      ia3_movie_job& j = P.jobs[who[m]->movie];
      j.n_rows[who[m]->sel] = items[m].n_rows;
      j.n_iter[who[m]->sel] = items[m].n_iter;
      if (items[m].rc) {
        std::lock_guard<std::mutex> lk(P.mu);
        if (!j.rc) { j.rc = items[m].rc; P.errs[(size_t)who[m]->movie] = items[m].err; }
      }
    }
    for (Image* im : grp) {
      ia3_movie_job& j = P.jobs[im->movie];
      j.t_fit_ms += dt / (double)grp.size();
      j.stamps[4] = ta - P.t0; j.stamps[5] = ta + dt - P.t0;
      release(im);
    }
  }
}

}  // namespace

extern "C" int ia3_process_movies(ia3_movie_job* jobs, int n_jobs, const ia3_movie_params* p) {
  int rc = ensure_init(); if (rc) return rc;
  if (n_jobs < 0 || (n_jobs > 0 && !jobs) || !p) return set_error(IA3_EINVAL, "null argument");
  if (p->frames < 1 || p->X < 1 || p->Y < 1 || p->Z < 1 || p->load_step < 1) return set_error(IA3_EINVAL, "bad movie shape");
  if (p->n_load < 1 || p->n_load > IA3_MOVIE_MAXCH || p->n_sel < 1 || p->n_sel > p->n_load || p->n_bleed < 0 || p->n_bleed > p->n_load)
    return set_error(IA3_EINVAL, "bad channel counts (%d loaded, %d selected, %d in the bleedthrough mix)", p->n_load, p->n_sel, p->n_bleed);
  for (int c = 0; c < p->n_load; ++c)
    if (p->load_start[c] < 0 || (long long)p->load_start[c] + (long long)(p->Z - 1) * p->load_step >= p->frames)
      return set_error(IA3_EINVAL, "channel %d: frames %d + k*%d (k < %d) fall outside a movie of %d frames", c, p->load_start[c], p->load_step, p->Z, p->frames);
  for (int s = 0; s < p->n_sel; ++s) {
    if (p->sel[s] < 0 || p->sel[s] >= p->n_load) return set_error(IA3_EINVAL, "selected channel %d is not a loaded one", s);
    for (int t = 0; t < s; ++t) if (p->sel[t] == p->sel[s]) return set_error(IA3_EINVAL, "channel selected twice");
  }
  for (int b = 0; b < p->n_bleed; ++b) {
    if (p->bleed_idx[b] < 0 || p->bleed_idx[b] >= p->n_load) return set_error(IA3_EINVAL, "bleedthrough channel %d is not a loaded one", b);
    for (int t = 0; t < b; ++t) if (p->bleed_idx[t] == p->bleed_idx[b]) return set_error(IA3_EINVAL, "channel twice in the bleedthrough mix");
  }
  if (p->n_bleed > 0 && !p->bleed_profile) return set_error(IA3_EINVAL, "no bleedthrough profile");
  if (p->drift_idx >= p->n_load) return set_error(IA3_EINVAL, "the bead channel is not a loaded one");
  if (p->drift_idx >= 0) {
    if (p->n_crops < 1 || p->n_crops > 8) return set_error(IA3_EINVAL, "1..8 drift crops");
    for (int i = 0; i < p->n_crops; ++i) {
      const int dims[3] = {p->Z, p->X, p->Y};
      for (int a = 0; a < 3; ++a)
        if (p->crops[i][a][0] < 0 || p->crops[i][a][1] > dims[a] || p->crops[i][a][0] >= p->crops[i][a][1])
          return set_error(IA3_EINVAL, "drift crop %d leaves the image", i);
    }
    if (p->precision_fold < 1 || p->min_good_drifts < 1) return set_error(IA3_EINVAL, "bad drift parameters");
  }
  if (p->normalize < 0 || p->normalize > 2 || (p->normalize && (!p->bg_edges || p->bg_n_edges < 2)))
    return set_error(IA3_EINVAL, "bad normalisation arguments");
  if (n_jobs == 0) return IA3_OK;
  for (int k = 0; k < n_jobs; ++k) {
    ia3_movie_job& j = jobs[k];
    j.rc = 0; j.drift_flag = 0;
    j.drift[0] = j.drift[1] = j.drift[2] = 0;
    j.t_upload_ms = j.t_correct_ms = j.t_fit_ms = 0;
    for (int q = 0; q < 6; ++q) j.stamps[q] = 0;
    for (int s = 0; s < IA3_MOVIE_MAXCH; ++s) j.n_rows[s] = j.n_seeds[s] = j.n_iter[s] = 0;
    if (p->fit_spots)
      for (int s = 0; s < p->n_sel; ++s)
        if (!j.rows[s] || j.capacity[s] < 1) return set_error(IA3_EINVAL, "movie %d, channel %d: no row buffer", k, s);
  }
  Pipe P;
  P.jobs = jobs; P.n_jobs = n_jobs; P.p = p; P.t0 = now_ms();
  P.raw.assign((size_t)n_jobs, nullptr);
  P.uploaded.assign((size_t)n_jobs, 0);
  P.errs.assign((size_t)n_jobs, std::string());
  P.group = p->fit_group_images > 0 ? p->fit_group_images : 12;
  if (P.group > ia3k::fit_max_fovs()) P.group = ia3k::fit_max_fovs();
  int NC = p->correct_threads > 0 ? p->correct_threads : 2;
  if (NC > 8) NC = 8;
  if (NC > n_jobs) NC = n_jobs;
  // the reference bead stack (and profiles) may still be in production on the caller's stream
  rc = stream_wait_spin(stream()); if (rc) return rc;
  bool any_measure = false;
  for (int k = 0; k < n_jobs; ++k) any_measure = any_measure || jobs[k].measure_drift;
  ia3_drift_ref* own_ref = nullptr;
  if (p->drift_idx >= 0 && any_measure) {
    if (p->drift_ref) P.dref = p->drift_ref;
    else {
      if (!p->ref_bead) return set_error(IA3_EINVAL, "no reference bead stack");
      if (p->ref_bead->Z != p->Z || p->ref_bead->X != p->X || p->ref_bead->Y != p->Y || p->ref_bead->dtype != IA3_U16)
        return set_error(IA3_EINVAL, "the reference bead stack must be a uint16 stack of the image size");
      rc = ia3_drift_ref_create(p->ref_bead, &p->crops[0][0][0], p->n_crops, &own_ref); if (rc) return rc;
      P.dref = own_ref;
    }
  }
  // The pipeline's depth is bounded (upload_ahead movies waiting, NC in correction, the fit queue three groups deep):
  // for a long batch, which will reach that depth, the blocks are taken from the driver now, not one by one in the middle of it
  if (n_jobs >= 16) {   // (a short batch would pay more for the allocations than it can lose to a stall)
    const int ahead = p->upload_ahead > 0 ? p->upload_ahead : 2;
    ws_reserve((size_t)p->frames * p->X * p->Y * sizeof(uint16_t), ahead + NC + 1);
    const int chain = p->n_load + p->n_bleed + 2 * p->n_sel;                    // stacks one correction holds at its widest
    const int queued = p->fit_spots ? 4 * P.group + NC * p->n_sel : NC * p->n_sel;
    ws_reserve((size_t)p->Z * p->X * p->Y * sizeof(uint16_t), NC * chain + queued);
  }
  ia3pipe::pool_run(NC + 2, [&]() {
    const int r = P.role.fetch_add(1);
    if (r == 0) fitter(P);
    else if (r == 1) uploader(P);
    else corrector(P);
  });
  if (own_ref) ia3_drift_ref_free(own_ref);
  for (int k = 0; k < n_jobs; ++k)
    if (jobs[k].rc) return set_error(jobs[k].rc, "movie %d: %s", k, P.errs[(size_t)k].c_str());
  return IA3_OK;
}
