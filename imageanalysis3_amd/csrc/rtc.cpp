// Kernels compiled at run time.  A few kernels of the library keep a whole z column of the stack in registers and are
// therefore built once per stack depth (gauss_col_*.hip, the spline prefilter's axis-0 pass in warp.hip); the library
// carries the depths of FOLD_DEPTHS, and a stack of another depth gets its kernels here: hiprtc on the kernel's own
// device-only text (files beside the library, csrc/), with the flags the Makefile uses, for the architecture of the
// current device.  The lowered names and the code object are kept in a cache file (IA3_RTC_CACHE, <library dir>/_rtc or
// /tmp/ia3_rtc_<uid>) under a hash of source, flags and compiler version, and in the process per device.  No process is
// started for this (a process that holds the GPU must not exec on the boxes this was developed on).
#include "ia3_rt.h"
#include <hip/hiprtc.h>
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

namespace {
struct Entry { bool ok = false; hipModule_t mod = nullptr; std::vector<hipFunction_t> fns; };
std::mutex g_mu;
std::map<unsigned long long, Entry> g_mods;
pid_t g_pid = 0;
int g_dev = -1;

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
  if (!dladdr((void*)&ia3rt::rtc_kernels, &info) || !info.dli_fname) return std::string();
  std::string p(info.dli_fname);
  const size_t k = p.rfind('/');
  return k == std::string::npos ? std::string(".") : p.substr(0, k);
}
std::string cache_dir(const std::string& lib) {
  const char* e = getenv("IA3_RTC_CACHE");
  const std::string cand[3] = {e ? std::string(e) : std::string(), lib + "/_rtc", std::string("/tmp/ia3_rtc_") + std::to_string((long)getuid())};
  for (const std::string& d : cand) {
    if (d.empty()) continue;
    (void)mkdir(d.c_str(), 0755);
    if (access(d.c_str(), W_OK) == 0) return d;
  }
  return std::string();
}
}  // namespace

namespace ia3rt {

bool rtc_kernels(const char* tag, const std::vector<std::string>& files, const std::string& preamble,
                 const std::vector<std::string>& names, std::vector<hipFunction_t>& fns) {
  std::lock_guard<std::mutex> lk(g_mu);
  int dev = -1;
  (void)hipGetDevice(&dev);
  if (g_pid != getpid() || g_dev != dev) { g_mods.clear(); g_pid = getpid(); g_dev = dev; }   // (modules of another process / device are not ours)
  std::string ident(tag);
  for (const std::string& f : files) ident += "|" + f;
  for (const std::string& n : names) ident += "|" + n;
  Entry& e = g_mods[fnv(preamble, fnv(ident))];
  if (!e.fns.empty() || e.mod) { fns = e.fns; return e.ok; }
  e.fns.assign(names.size(), nullptr);   // tried: a failure is not tried again in this process
  const char* off = getenv("IA3_RTC");
  if (off && atoi(off) == 0) return false;
  const std::string lib = lib_dir();
  if (lib.empty()) return false;
  std::string src = preamble;
  for (const std::string& f : files) {
    std::string text;
    if (!read_file(lib + "/csrc/" + f, text)) return false;
    const size_t po = text.find("#pragma once");
    if (po != std::string::npos) text.replace(po, 12, "");
    src += text + "\n";
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
  std::string arch(prop.gcnArchName);
  { const size_t c = arch.find(':'); if (c != std::string::npos) arch.resize(c); }
  const std::string a_opt = "--offload-arch=" + arch;
  const char* opts[] = {a_opt.c_str(), "-O3", "-std=c++17", "-ffp-contract=off"};
  int major = 0, minor = 0;
  (void)hiprtcVersion(&major, &minor);
  const unsigned long long key = fnv(ident + a_opt + std::to_string(major) + "." + std::to_string(minor), fnv(src));
  const std::string cdir = cache_dir(lib);
  char fname[96];
  snprintf(fname, sizeof(fname), "/%s_%016llx.bin", tag, key);
  std::string blob;   // [lowered name]\0 ... [code object]
  bool cached = !cdir.empty() && read_file(cdir + fname, blob) && blob.size() > 16;
  for (int attempt = 0; attempt < 2; ++attempt) {   // (second round: a cache file that would not load is thrown away and made again)
  if (!cached) {
    if (!getenv("IA3_RTC_QUIET"))   // the one place the library speaks unasked: the caller would otherwise sit through a silent half minute
      fprintf(stderr, "ia3: compiling %s for this stack depth (once per depth, dtype and machine; kept in %s)\n", names[0].c_str(),
              cdir.empty() ? "memory only" : cdir.c_str());
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "ia3_rtc.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return false;
    bool ok = true;
    for (const std::string& n : names) ok = ok && hiprtcAddNameExpression(prog, n.c_str()) == HIPRTC_SUCCESS;
    ok = ok && hiprtcCompileProgram(prog, 4, opts) == HIPRTC_SUCCESS;
    blob.clear();
    for (size_t i = 0; ok && i < names.size(); ++i) {
      const char* low = nullptr;
      ok = hiprtcGetLoweredName(prog, names[i].c_str(), &low) == HIPRTC_SUCCESS && low;
      if (ok) { blob.append(low); blob.push_back('\0'); }
    }
    size_t cs = 0;
    ok = ok && hiprtcGetCodeSize(prog, &cs) == HIPRTC_SUCCESS && cs > 0;
    if (ok) {
      const size_t at = blob.size();
      blob.resize(at + cs);
      ok = hiprtcGetCode(prog, &blob[at]) == HIPRTC_SUCCESS;
    } else if (getenv("IA3_RTC_VERBOSE")) {
      size_t ls = 0;
      (void)hiprtcGetProgramLogSize(prog, &ls);
      std::string log(ls, '\0');
      if (ls) (void)hiprtcGetProgramLog(prog, &log[0]);
      fprintf(stderr, "ia3: run-time compile (%s) failed:\n%s\n", ident.c_str(), log.c_str());
    }
    (void)hiprtcDestroyProgram(&prog);
    if (!ok) return false;
    if (!cdir.empty()) {   // written under a private name, then moved into place (other processes may compile the same kernels)
      const std::string tmp = cdir + fname + "." + std::to_string((long)getpid());
      FILE* f = fopen(tmp.c_str(), "wb");
      if (f) {
        const bool w = fwrite(blob.data(), 1, blob.size(), f) == blob.size();
        fclose(f);
        if (!w || rename(tmp.c_str(), (cdir + fname).c_str()) != 0) (void)unlink(tmp.c_str());
      }
    }
  }
  std::vector<const char*> low(names.size(), "");
  size_t at = 0;
  for (size_t i = 0; i < names.size() && at < blob.size(); ++i) {
    low[i] = blob.c_str() + at;
    at += strlen(low[i]) + 1;
  }
  bool loaded = at < blob.size() && hipModuleLoadData(&e.mod, blob.data() + at) == hipSuccess;
  for (size_t i = 0; loaded && i < names.size(); ++i) loaded = hipModuleGetFunction(&e.fns[i], e.mod, low[i]) == hipSuccess;
  if (loaded) break;
  (void)hipGetLastError();
  if (e.mod) { (void)hipModuleUnload(e.mod); e.mod = nullptr; }
  e.fns.assign(names.size(), nullptr);
  if (!cached) return false;
  (void)unlink((cdir + fname).c_str());
  cached = false;
  }
  if (!e.mod) return false;
  e.ok = true;
  fns = e.fns;
  return true;
}

}  // namespace ia3rt
