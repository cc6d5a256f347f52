// Depth dispatch of the column-in-registers axis-0 kernels: one translation unit per (dtype, depth) is built from
// gauss_col_one.hip for every depth of IA3_FOLD_DEPTHS (Makefile: FOLD_DEPTHS, default 25 30 33 35 40 45 50 60 — the
// reference takes any `single_im_size`, io_tools/load.py:166-180; its default depth is 30, __init__.py:8-20).  Any
// other depth from 16 to 64 planes gets its kernels from the run-time compiler (hiprtc, below) the first time it is
// asked for — the same kernel text, the same flags; ~35 s once per depth, dtype and machine, then from a cache file —
// and only when that is not possible (IA3_RTC=0, no hiprtc, sources not beside the library) the sliding-window kernels
// (same results, +0.65 ms per 2048 x 2048 x 50 stack).
#include "ia3_gauss.h"
#include <hip/hiprtc.h>
#include <dlfcn.h>
#include <sys/stat.h>
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

bool read_file(const std::string& path, std::string& out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char buf[65536];
  size_t n;
  out.clear();
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out.append(buf, n);
  fclose(f);
  return true;
}
unsigned long long fnv(const std::string& s, unsigned long long h = 1469598103934665603ull) {
  for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
  return h;
}
std::string lib_dir() {
  Dl_info info;
  if (!dladdr((void*)&ia3g::folded_axis0_f32, &info) || !info.dli_fname) return std::string();
  std::string p(info.dli_fname);
  const size_t k = p.rfind('/');
  return k == std::string::npos ? std::string(".") : p.substr(0, k);
}
std::string cache_dir(const std::string& lib) {
  const char* e = getenv("IA3_RTC_CACHE");
  std::string cand[3] = {e ? std::string(e) : std::string(), lib + "/_rtc", std::string("/tmp/ia3_rtc_") + std::to_string((long)getuid())};
  for (const std::string& d : cand) {
    if (d.empty()) continue;
    (void)mkdir(d.c_str(), 0755);
    if (access(d.c_str(), W_OK) == 0) return d;
  }
  return std::string();
}

// code object + the two kernels of (dtype, depth); nullptr when the run-time path is not available (the caller then
// reports FOLD_NOT_COVERED and the sliding-window kernels run).  Called with g_rtc_mu held.
const RtcDepth* rtc_depth_locked(bool f32, int Z) {
  int dev = -1;
  (void)hipGetDevice(&dev);
  if (g_rtc_pid != getpid() || g_rtc_dev != dev) { g_rtc.clear(); g_rtc_rows.clear(); g_rtc_pid = getpid(); g_rtc_dev = dev; }
  RtcDepth& d = g_rtc[Z * 2 + (f32 ? 1 : 0)];
  if (d.tried) return d.axis0 ? &d : nullptr;
  d.tried = true;
  const char* off = getenv("IA3_RTC");
  if ((off && atoi(off) == 0) || Z < 16 || Z > 64) return nullptr;
  const std::string lib = lib_dir();
  std::string dev_h, kern;
  if (lib.empty() || !read_file(lib + "/csrc/ia3_gauss_dev.h", dev_h) || !read_file(lib + "/csrc/gauss_col_kernel.inc", kern)) return nullptr;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return nullptr;
  std::string arch(prop.gcnArchName);
  { const size_t c = arch.find(':'); if (c != std::string::npos) arch.resize(c); }
  char pre[512];
  snprintf(pre, sizeof(pre),
           "#define IA3_MODE_REFLECT %d\n#define IA3_MODE_NEAREST %d\n#define IA3_MODE_CONSTANT %d\n"
           "namespace ia3k { constexpr int DOG_PAIR_ZGROUPS = %d; }\n",
           (int)IA3_MODE_REFLECT, (int)IA3_MODE_NEAREST, (int)IA3_MODE_CONSTANT, (int)ia3k::DOG_PAIR_ZGROUPS);
  { const size_t po = dev_h.find("#pragma once"); if (po != std::string::npos) dev_h.replace(po, 12, ""); }
  const std::string src = std::string(pre) + dev_h + "\n" + kern + "\n";
  const char* tname = f32 ? "float" : "unsigned short";
  char n0[160], n1[160];
  snprintf(n0, sizeof(n0), "ia3colk::gauss_axis0_folded<%s, %d, %d, 0>", tname, Z, RTC_R);
  snprintf(n1, sizeof(n1), "ia3colk::gauss_axis0_folded<%s, %d, %d, %d>", tname, Z, RTC_R, RTC_RF);
  const std::string a_opt = "--offload-arch=" + arch;
  const char* opts[] = {a_opt.c_str(), "-O3", "-std=c++17", "-ffp-contract=off"};
  int rv = 0;
  (void)hiprtcVersion(&rv, &rv);
  const unsigned long long key = fnv(std::string(n0) + n1 + a_opt + std::to_string(rv), fnv(src));
  const std::string cdir = cache_dir(lib);
  char fname[64];
  snprintf(fname, sizeof(fname), "/col_%016llx.bin", key);
  std::string blob;   // [lowered name 0]\0[lowered name 1]\0[code object]
  bool cached = !cdir.empty() && read_file(cdir + fname, blob) && blob.size() > 16;
  if (!cached) {
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "gauss_col_rtc.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return nullptr;
    bool ok = hiprtcAddNameExpression(prog, n0) == HIPRTC_SUCCESS && hiprtcAddNameExpression(prog, n1) == HIPRTC_SUCCESS &&
              hiprtcCompileProgram(prog, 4, opts) == HIPRTC_SUCCESS;
    const char *l0 = nullptr, *l1 = nullptr;
    size_t cs = 0;
    ok = ok && hiprtcGetLoweredName(prog, n0, &l0) == HIPRTC_SUCCESS && hiprtcGetLoweredName(prog, n1, &l1) == HIPRTC_SUCCESS &&
         hiprtcGetCodeSize(prog, &cs) == HIPRTC_SUCCESS && cs > 0;
    if (ok) {
      blob.assign(l0); blob.push_back('\0'); blob.append(l1); blob.push_back('\0');
      const size_t at = blob.size();
      blob.resize(at + cs);
      ok = hiprtcGetCode(prog, &blob[at]) == HIPRTC_SUCCESS;
    } else if (getenv("IA3_RTC_VERBOSE")) {
      size_t ls = 0;
      (void)hiprtcGetProgramLogSize(prog, &ls);
      std::string log(ls, '\0');
      if (ls) (void)hiprtcGetProgramLog(prog, &log[0]);
      fprintf(stderr, "ia3: run-time compile of the depth-%d column kernel failed:\n%s\n", Z, log.c_str());
    }
    (void)hiprtcDestroyProgram(&prog);
    if (!ok) return nullptr;
    if (!cdir.empty()) {   // written under a private name, then moved into place (other processes compile the same depth)
      const std::string tmp = cdir + fname + "." + std::to_string((long)getpid());
      FILE* f = fopen(tmp.c_str(), "wb");
      if (f) {
        const bool w = fwrite(blob.data(), 1, blob.size(), f) == blob.size();
        fclose(f);
        if (!w || rename(tmp.c_str(), (cdir + fname).c_str()) != 0) (void)unlink(tmp.c_str());
      }
    }
  }
  const char* l0 = blob.c_str();
  const size_t len0 = strlen(l0);
  if (len0 + 2 >= blob.size()) return nullptr;
  const char* l1 = l0 + len0 + 1;
  const size_t len1 = strlen(l1);
  if (len0 + len1 + 2 >= blob.size()) return nullptr;
  const char* code = l1 + len1 + 1;
  if (hipModuleLoadData(&d.mod, code) != hipSuccess) { (void)hipGetLastError(); d.mod = nullptr; return nullptr; }
  if (hipModuleGetFunction(&d.axis0, d.mod, l0) != hipSuccess || hipModuleGetFunction(&d.pair, d.mod, l1) != hipSuccess) {
    (void)hipGetLastError();
    d.axis0 = d.pair = nullptr;
    return nullptr;
  }
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
