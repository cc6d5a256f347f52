// Internal runtime of libia3.so: lazy per-process HIP context, error reporting, cached device
// scratch, stack handles.  One HIP stream per process (SURVEY.md §8b "Threading").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include "../../include/ia3.h"

struct ia3_stack {
  void* d;        // device pointer
  int dtype, Z, X, Y;
  bool owned;
  size_t bytes;
  void* home = nullptr;   // hipStream_t of the thread that allocated it (owned stacks): see ia3_stack_free
};

namespace ia3rt {

int set_error(int code, const char* fmt, ...);
int ensure_init();
hipStream_t stream();
int num_cus();  // compute units of the selected device
// pinned, device-mapped host block of the calling thread (>= bytes, zero-initialised when it is made) and a stream wait
// that spins on an event: small results reach the host without a copy into pageable memory and a sleeping synchronise
int host_mailbox(size_t bytes, void** host, void** dev);
int stream_wait_spin(hipStream_t st);
// One step of a host-side polling loop: a CPU relax hint for the first ~100-200 us (what the per-FOV path waits for is
// microseconds away), afterwards the core is given up between looks (a group fit runs for tens of milliseconds and up to
// 17 library threads may be waiting at once: they must not take the cores of the callers and of the upload copies).
struct SpinWait {
  unsigned long long n = 0;
  void relax();
};
int set_upload_threads(int n);   // IA3_TUNE_UPLOAD_THREADS
inline size_t esize(int dtype) { return dtype == IA3_U16 ? 2 : 4; }

// Cached device scratch: get(bytes) returns a buffer that stays valid until put(); buffers are
// reused across calls (hipMalloc of ~GB blocks costs milliseconds).
void* ws_get(size_t bytes);
void ws_put(void* p);
void ws_release_all();
// runtime.cpp: a four-float device slot beside a buffer made by ia3_buffer_upload (constant over a run), see there
float* const_note(const void* buf, bool* ready, bool* fill);
void const_note_filled(const void* buf, bool ok);
// rtc.cpp: kernels compiled at run time from device-only files beside the library (csrc/<files>, preceded by `preamble`);
// fns[i] is the kernel of name expression names[i].  false: not available (IA3_RTC=0, no sources, compile error)
bool rtc_kernels(const char* tag, const std::vector<std::string>& files, const std::string& preamble,
                 const std::vector<std::string>& names, std::vector<hipFunction_t>& fns);
void ws_reserve(size_t bytes, int count);   // at least `count` cached blocks that fit a request of `bytes` (best effort)

struct Scratch {  // RAII
  void* p;
  explicit Scratch(size_t bytes) : p(ws_get(bytes)) {}
  ~Scratch() { if (p) ws_put(p); }
  Scratch(const Scratch&) = delete;
  Scratch& operator=(const Scratch&) = delete;
  template <class T> T* as() const { return (T*)p; }
};

#define IA3_HIP(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess)                                                                  \
      return ia3rt::set_error(IA3_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                              __FILE__, __LINE__);                                         \
  } while (0)

#define IA3_KCHECK()                                                                       \
  do {                                                                                     \
    hipError_t _e = hipGetLastError();                                                     \
    if (_e != hipSuccess)                                                                  \
      return ia3rt::set_error(IA3_EHIP, "kernel launch failed: %s (%s:%d)",                \
                              hipGetErrorString(_e), __FILE__, __LINE__);                  \
  } while (0)

// Work queued while an AuxScope is alive goes to a second stream that first waits for everything queued on the main
// stream so far; aux_join() makes the main stream wait for it.  Used to run independent kernels side by side (a
// VALU-bound one next to a memory/LDS-bound one).  Scratch blocks released inside the scope are recycled after the join.
struct AuxScope {
  bool ok;
  AuxScope();
  ~AuxScope();
};
// back onto the auxiliary stream behind an earlier AuxScope of the same fork (no new fork point on the main stream)
struct AuxResume {
  bool ok;
  AuxResume();
  ~AuxResume();
};
int aux_join();
// see runtime.cpp: blocks handed to ws_put inside the scope go back to the cache later (ws_put_deferred_now / scope end)
struct PutDefer {
  bool outer;
  PutDefer();
  ~PutDefer();
};
void ws_put_deferred_now();
// developer aid: with IA3_DEBUG_TIMES set, prints a monotonic time stamp (us) and the label to stderr
void dbg_stamp(const char* what);

// Optional per-kernel timing with HIP events on the library stream (ia3_profile_*): bench.py
// derives roofline.achieved from these, rocprofv3 must agree.
struct ProfScope {
  int slot;
  explicit ProfScope(const char* name);
  ~ProfScope();
};

// scipy.ndimage._gaussian_kernel1d(order 0) incl. NumPy's pairwise summation order
void gaussian_taps(double sigma, double truncate, std::vector<double>& w, int& radius);

}  // namespace ia3rt

// ---- stage entry points implemented in the .hip files (device pointers, library stream) ------
namespace ia3k { void set_dft_valu(int on); void set_fft_c2c(int on); void set_seed_dense(int on);
void set_seed_strips(int on); void set_fit_nblist(int cap); void set_fit_fuse(int on); void set_fit_waves(int n); void set_fit_maxfev(int n); void set_fit_merge(int on); void set_warp_onepass(int v); void set_fit_kdq(int cap); void set_fit_memo(int on); int set_fit_waitbound(int polls); }   // fft_align.hip: test knob, see IA3_TUNE_DFT_VALU
namespace ia3k {
// separable Gaussian along all three axes: src -> dst, tmp is a same-size scratch stack.  axes: bit 0 = the axis-0
// pass (src -> dst), bit 1 = the axis-1 and axis-2 passes (dst -> tmp -> dst); radius <= 3 runs fused (axes == 3 only).
int gaussian3d(const void* src, int dtype, int Z, int X, int Y, const double* w, int radius, int mode,
               void* dst, void* tmp, int axes = 3);
// out = im - low ; out[low > im] = 0   (correction_tools/filter.py:17-18)
int highpass_combine(const void* im, const void* low, int dtype, size_t n, void* out);
// DoG pair of the seed detector: short filter -> dst_front (complete), axis-0 pass of the long filter -> dst_zp (gauss.hip)
// tmax (optional): per plane, 16-row step and y tile (dog_pair_tiles) the largest value of dst_front, for the detector
// smin / sabs (optional, dog_pair_strips(X, Y) floats each, only when that is non-zero): smallest value / largest magnitude
// of dst_zp per group of planes (group of plane z = z * DOG_PAIR_ZGROUPS / Z), row and 32-column strip
constexpr int DOG_PAIR_ZGROUPS = 5;
size_t dog_pair_strips(int X, int Y);
int gauss_dog_pair(const void* src, int dtype, int Z, int X, int Y, const double* wf, int rf, const double* wb, int rb,
                   void* dst_front, void* dst_zp, void* tmp, int* forked, float* tmax = nullptr, float* smin = nullptr,
                   float* sabs = nullptr);
void dog_pair_tiles(int X, int Y, int* ty, int* ntile, size_t* count);
// get_seeds on a resident stack (seed.hip)
struct SeedOut {
  std::vector<double> zxyh;  // n x 4 [z,x,y,h], brightest first
  double th_used;
};
int dog_seed(const ia3_stack* im, const ia3_seed_params& p, SeedOut& out);
// same, with the seed list left on the device when the device-side finish applies (<= 32768 candidates, stack no
// larger than 256 x 4096 x 4096): d_zxy = n x 3 float64 centres, d_h = n float64 heights inside `hold`; otherwise
// on_device == false and `host` carries the list.
struct SeedDev {
  bool on_device = false;
  int n = 0;
  double th_used = 0;
  const double* d_zxy = nullptr;
  const double* d_h = nullptr;
  void* hold = nullptr;     // scratch block that owns d_zxy / d_h; release with ws_put
  // called (if set) when the device-side finish has been QUEUED and before the host waits for its count: d_zxy is where
  // the seeds will be, *d_count how many survive (before the max_num_seeds cut), both valid once the stream gets there.
  // ia3_fit_fov_dev queues the fitter's set-up kernels from here, behind the seed stage.
  void (*ahead)(void* ctx, const double* d_zxy, const unsigned* d_count, int max_num_seeds) = nullptr;
  void* ahead_ctx = nullptr;
  SeedOut host;
  ~SeedDev() { if (hold) ia3rt::ws_put(hold); }
};
int dog_seed_dev(const ia3_stack* im, const ia3_seed_params& p, SeedDev& out);
// fitter from centres that are already resident (n x 3 float64)
int fit_create_dev(const ia3_stack* im, const double* d_centers_zxy, int n, const ia3_fit_params* p, ia3_fitter** out);
// the same before the host knows the count (see fit.hip): capacity seeds laid out, count = min(*d_count, n_cut if > 0)
int fit_create_ahead(const ia3_stack* im, const double* d_centers_zxy, int capacity, const unsigned* d_count, int n_cut,
                     const ia3_fit_params* p, ia3_fitter** out);
int fit_set_count(ia3_fitter* f, int n);
// one fitter over several resident fields of view (same shape and dtype; at most fit_max_fovs()): seeds of field k =
// n_seeds[k] x 3 float64 at d_centers_zxy[k].  fit_fov_results (after ia3_fit_results(_ex)): per field its sweep count and
// its fits / evaluations / voxel evaluations; fit_fov_starts: n_fov + 1 row offsets of the fields in the row table
int fit_max_fovs();
int fit_create_multi(const ia3_stack* const* ims, const double* const* d_centers_zxy, const int* n_seeds, int n_fov,
                     const ia3_fit_params* p, ia3_fitter** out);
int fit_fov_results(ia3_fitter* f, int* n_iter, long long* counters3);
const int* fit_fov_starts(const ia3_fitter* f);
// drift of an image against a reference bead image whose crop spectra are kept (fft_align.hip): crops = n x 6 ints
// [z0, z1, x0, x1, y0, y1]; eager = all spectra now (a reference shared by several threads), else each when first needed
struct DriftRef;
int drift_ref_create(const ia3_stack* ref, const int* crops, int n_crops, bool eager, DriftRef** out);
void drift_ref_free(DriftRef* r);
int drift_crops(const ia3_stack* src, DriftRef* ref, int first, int count, int upsample, int normalization, double* shifts);
// fits run / model evaluations / voxel evaluations / shader cycles in dependency waits / wave cycles of a fitter, as of its
// last ia3_fit_results(_ex)
void fit_host_counters(const ia3_fitter* f, long long out[5]);
}  // namespace ia3k
