// Depth dispatch of the column-in-registers axis-0 kernels: one translation unit per (dtype, depth) is built from
// gauss_col_one.hip for every depth of IA3_FOLD_DEPTHS (Makefile: FOLD_DEPTHS, default 25 30 33 35 40 45 50 60 — the
// reference takes any `single_im_size`, io_tools/load.py:166-180; its default depth is 30, __init__.py:8-20).  Any
// other depth from 16 to 64 planes gets its kernels from the run-time compiler (hiprtc, below) the first time it is
// asked for — the same kernel text, the same flags; ~35 s once per depth, dtype and machine, then from a cache file —
// and only when that is not possible (IA3_RTC=0, no hiprtc, sources not beside the library) the sliding-window kernels
// (same results, +0.65 ms per 2048 x 2048 x 50 stack).
#include "ia3_gauss.h"
#include <unistd.h>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace ia3g {

#define IA3_DECL(ZZ)                                                                                                        \
  int folded_axis0_f32_z##ZZ(const float*, size_t, const Taps&, int, float*, hipStream_t, int);                             \
  int folded_axis0_u16_z##ZZ(const uint16_t*, size_t, const Taps&, int, uint16_t*, hipStream_t, int);                       \
  int folded_pair_f32_z##ZZ(const float*, size_t, const Taps&, float*, const Taps&, float*, hipStream_t, int, float*, float*, int);   \
  int folded_pair_u16_z##ZZ(const uint16_t*, size_t, const Taps&, uint16_t*, const Taps&, uint16_t*, hipStream_t, int, float*, float*, int);
IA3_FOLD_DEPTHS(IA3_DECL)
#undef IA3_DECL

}  // namespace ia3g

// ---- run-time compiled depths -------------------------------------------------------------------------------------
namespace {
using ia3g::Taps;
struct RtcDepth { bool tried = false; hipModule_t mod = nullptr; hipFunction_t axis0 = nullptr, pair = nullptr; };
struct RtcRows { int Z, mode; std::vector<double> w; double* d; };
std::mutex g_rtc_mu;
std::map<int, RtcDepth> g_rtc;          // key: depth * 2 + (float32 ? 1 : 0)
std::vector<RtcRows> g_rtc_rows;
pid_t g_rtc_pid = 0;
int g_rtc_dev = -1;
constexpr int RTC_R = 30, RTC_RF = 3;   // the radii the translation units are built for (gauss_col.inc)

// the two kernels of (dtype, depth); nullptr when the run-time path is not available (the caller then reports
// FOLD_NOT_COVERED and the sliding-window kernels run).  Called with g_rtc_mu held.
const RtcDepth* rtc_depth_locked(bool f32, int Z) {
  int dev = -1;
  (void)hipGetDevice(&dev);
  if (g_rtc_pid != getpid() || g_rtc_dev != dev) { g_rtc.clear(); g_rtc_rows.clear(); g_rtc_pid = getpid(); g_rtc_dev = dev; }
  RtcDepth& d = g_rtc[Z * 2 + (f32 ? 1 : 0)];
  if (d.tried) return d.axis0 ? &d : nullptr;
  d.tried = true;
  if (Z < 16 || Z > 64) return nullptr;
  char pre[512];
  snprintf(pre, sizeof(pre),
           "#define IA3_MODE_REFLECT %d\n#define IA3_MODE_NEAREST %d\n#define IA3_MODE_CONSTANT %d\n"
           "namespace ia3k { constexpr int DOG_PAIR_ZGROUPS = %d; }\n",
           (int)IA3_MODE_REFLECT, (int)IA3_MODE_NEAREST, (int)IA3_MODE_CONSTANT, (int)ia3k::DOG_PAIR_ZGROUPS);
  const char* tname = f32 ? "float" : "unsigned short";
  char n0[160], n1[160];
  snprintf(n0, sizeof(n0), "ia3colk::gauss_axis0_folded<%s, %d, %d, 0>", tname, Z, RTC_R);
  snprintf(n1, sizeof(n1), "ia3colk::gauss_axis0_folded<%s, %d, %d, %d>", tname, Z, RTC_R, RTC_RF);
  std::vector<hipFunction_t> fns;
  if (!ia3rt::rtc_kernels("col", {"ia3_gauss_dev.h", "gauss_col_kernel.inc"}, pre, {n0, n1}, fns)) return nullptr;
  d.axis0 = fns[0]; d.pair = fns[1];
  return &d;
}

// the folded weight rows of a depth (gauss_col.inc: folded_rows<Z, R>, the same sums in the same order), on the device
const double* rtc_rows_locked(int Z, const Taps& t, int mode) {
  constexpr int R = RTC_R;
  for (auto& e : g_rtc_rows)
    if (e.Z == Z && e.mode == mode && std::memcmp(e.w.data(), t.w, (R + 1) * sizeof(double)) == 0) return e.d;
  std::vector<double> rows;
  for (int z = 0; z < (Z + 1) / 2; ++z) {
    const int lo = z - R > 0 ? z - R : 0, hi = z + R < Z - 1 ? z + R : Z - 1;
    for (int p = lo; p <= hi; ++p) {
      double acc = 0.0;
      for (int j = -R; j <= R; ++j)
        if (ia3g::border_idx(z + j, Z, mode) == p) acc += t.w[j < 0 ? -j : j];
      rows.push_back(acc);
    }
  }
  rows.resize((rows.size() + 15) / 16 * 16, 0.0);
  double* d = nullptr;
  if (hipMalloc((void**)&d, rows.size() * sizeof(double)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
  if (g_rtc_rows.size() >= 16) g_rtc_rows.erase(g_rtc_rows.begin());   // (the evicted table stays allocated: launches may be in flight)
  g_rtc_rows.push_back(RtcRows{Z, mode, std::vector<double>(t.w, t.w + R + 1), d});
  return d;
}

int rtc_launch(bool f32, int Z, const void* src, size_t plane, const Taps& t, int mode, void* dst, const Taps& ft, void* fdst, hipStream_t s,
               int cert, float* smin, float* sabs, int Y, bool pair) {
  hipFunction_t fn;
  const double* wf;
  {
    std::lock_guard<std::mutex> lk(g_rtc_mu);
    const RtcDepth* d = rtc_depth_locked(f32, Z);
    if (!d) return ia3g::FOLD_NOT_COVERED;
    wf = rtc_rows_locked(Z, t, mode);
    if (!wf) return ia3rt::set_error(IA3_ENOMEM, "folded weight table");
    fn = pair ? d->pair : d->axis0;
  }
  // (const T* in, T* out, size_t plane, const double* wf, Taps taps, int mode, int cert, T* fout, Taps ftaps, float* smin, float* sabs, int Y)
  Taps ta = t, tb = ft;
  void* args[] = {(void*)&src, (void*)&dst, (void*)&plane, (void*)&wf, (void*)&ta, (void*)&mode, (void*)&cert, (void*)&fdst, (void*)&tb,
                  (void*)&smin, (void*)&sabs, (void*)&Y};
  if (hipModuleLaunchKernel(fn, (unsigned)((plane + 255) / 256), 1, 1, 256, 1, 1, 0, s, args, nullptr) != hipSuccess)
    return ia3rt::set_error(IA3_EHIP, "launch of the run-time compiled column kernel (depth %d) failed: %s", Z, hipGetErrorString(hipGetLastError()));
  return 0;
}
}  // namespace

namespace ia3g {

int folded_axis0_f32(const float* src, int Z, size_t plane, const Taps& t, int mode, float* dst, hipStream_t s, int cert) {
  switch (Z) {
#define IA3_FOLD_CASE(ZZ) case ZZ: return folded_axis0_f32_z##ZZ(src, plane, t, mode, dst, s, cert);
    IA3_FOLD_DEPTHS(IA3_FOLD_CASE)
#undef IA3_FOLD_CASE
    default: return rtc_launch(true, Z, src, plane, t, mode, dst, t, nullptr, s, cert, nullptr, nullptr, 0, false);
  }
}
int folded_axis0_u16(const uint16_t* src, int Z, size_t plane, const Taps& t, int mode, uint16_t* dst, hipStream_t s, int cert) {
  switch (Z) {
#define IA3_FOLD_CASE(ZZ) case ZZ: return folded_axis0_u16_z##ZZ(src, plane, t, mode, dst, s, cert);
    IA3_FOLD_DEPTHS(IA3_FOLD_CASE)
#undef IA3_FOLD_CASE
    default: return rtc_launch(false, Z, src, plane, t, mode, dst, t, nullptr, s, cert, nullptr, nullptr, 0, false);
  }
}
int folded_pair_f32(const float* src, int Z, size_t plane, const Taps& bt, float* dst, const Taps& ft, float* fdst, hipStream_t s, int cert,
                    float* smin, float* sabs, int Y) {
  switch (Z) {
#define IA3_FOLD_CASE(ZZ) case ZZ: return folded_pair_f32_z##ZZ(src, plane, bt, dst, ft, fdst, s, cert, smin, sabs, Y);
    IA3_FOLD_DEPTHS(IA3_FOLD_CASE)
#undef IA3_FOLD_CASE
    default: return rtc_launch(true, Z, src, plane, bt, IA3_MODE_REFLECT, dst, ft, fdst, s, cert, smin, sabs, Y, true);
  }
}
int folded_pair_u16(const uint16_t* src, int Z, size_t plane, const Taps& bt, uint16_t* dst, const Taps& ft, uint16_t* fdst, hipStream_t s,
                    int cert, float* smin, float* sabs, int Y) {
  switch (Z) {
#define IA3_FOLD_CASE(ZZ) case ZZ: return folded_pair_u16_z##ZZ(src, plane, bt, dst, ft, fdst, s, cert, smin, sabs, Y);
    IA3_FOLD_DEPTHS(IA3_FOLD_CASE)
#undef IA3_FOLD_CASE
    default: return rtc_launch(false, Z, src, plane, bt, IA3_MODE_REFLECT, dst, ft, fdst, s, cert, smin, sabs, Y, true);
  }
}

// 0: no column kernel for this depth (the sliding-window kernels run), 1: a translation unit of the library,
// 2: compiled at run time (now, or taken from the cache)
int column_kernel_source(bool f32, int Z) {
  if (fold_depth(Z)) return 1;
  std::lock_guard<std::mutex> lk(g_rtc_mu);
  return rtc_depth_locked(f32, Z) ? 2 : 0;
}

}  // namespace ia3g
