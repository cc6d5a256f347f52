// libia3.so runtime: context, errors, scratch cache, stack handles (host side of the C ABI).
#include "ia3_rt.h"
#include <time.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <unistd.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <map>
#include <mutex>
#include <atomic>
#include <thread>
#include <sched.h>

namespace ia3rt {

static thread_local char g_err[1024] = "";
static int g_device = -1;
static hipStream_t g_stream = nullptr;    // stream of the thread that initialised the library (ia3_stream())
static pid_t g_pid = 0;
static int g_cus = 256;
static std::mutex g_mu;

// Every host thread that calls into the library gets its own streams: the thread that initialised the library uses
// g_stream, any other thread a stream created on first use.  Independent images can therefore be processed side by
// side from a thread pool (one FOV per thread): the long tail of one image's fit kernel (a few fits that run to
// maxfev) overlaps the other images' work.
struct ThreadCtx {
  pid_t pid = 0;
  hipStream_t main = nullptr, aux = nullptr, cur = nullptr;   // cur: what stream() hands out (aux inside an AuxScope)
  bool defer_puts = false;   // PutDefer: ws_put only queues
  hipEvent_t fork = nullptr, join = nullptr;
  std::vector<void*> deferred;   // scratch blocks released inside an AuxScope
  void* mail_host = nullptr;     // pinned, device-mapped host block of this thread (host_mailbox)
  void* mail_dev = nullptr;
  size_t mail_bytes = 0;
  hipEvent_t mail_ev = nullptr;  // for wait_event_spin
  ~ThreadCtx();
};
static thread_local ThreadCtx t_ctx;

// Scratch cache with stream-ordered reuse: a released block remembers the stream it was last used on and an event
// recorded there at release time; a different stream that picks it up waits for that event first.
struct WsEntry { void* p; size_t bytes; bool busy; hipStream_t last; hipEvent_t ev; };
static std::vector<WsEntry> g_ws;

// A host thread that ends hands its streams, events and mailbox back: callers that start a pool of threads per batch
// (ThreadPoolExecutor around fit_fov_image) would otherwise leave two streams per dead thread multiplexed on the 16
// hardware queues next to the live ones.  The process's first stream (g_stream) stays; after a fork nothing here is ours.
// the main streams of the host threads that are alive (under g_mu): a stack remembers the stream of the thread that made
// it (ia3_stack_free), and that thread may be gone
struct ConstNote { float* slot = nullptr; int state = 0; };
static std::map<void*, ConstNote> g_const_bufs;   // buffers of ia3_buffer_upload and their notes (see there); under g_mu
static std::vector<hipStream_t> g_live_streams;
static bool stream_alive_locked(hipStream_t s) {
  for (hipStream_t t : g_live_streams) if (t == s) return true;
  return false;
}
ThreadCtx::~ThreadCtx() {
  if (!main || pid != getpid() || main == g_stream) return;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = 0; i < g_live_streams.size(); ++i)
      if (g_live_streams[i] == main) { g_live_streams[i] = g_live_streams.back(); g_live_streams.pop_back(); break; }
  }
  (void)hipStreamSynchronize(main);
  if (aux) { (void)hipStreamSynchronize(aux); (void)hipStreamDestroy(aux); }
  (void)hipStreamDestroy(main);
  if (fork) (void)hipEventDestroy(fork);
  if (join) (void)hipEventDestroy(join);
  if (mail_ev) (void)hipEventDestroy(mail_ev);
  if (mail_host) (void)hipHostFree(mail_host);
}

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

static int do_init(int device) {
  // HIP state does not survive fork(): a child that inherits g_pid != getpid() starts over.
  if (g_stream && g_pid == getpid() && (device < 0 || device == g_device)) return IA3_OK;
  if (g_pid != getpid()) { g_stream = nullptr; g_ws.clear(); g_live_streams.clear(); g_const_bufs.clear(); g_device = -1; }
  // Every host thread drives two streams (ThreadCtx); the runtime multiplexes streams onto 4 hardware queues by
  // default, where a long-running fit kernel holds back unrelated work queued behind it.  Ask for more queues unless
  // the user chose a number (only effective when this is the first HIP call of the process).
  setenv("GPU_MAX_HW_QUEUES", "16", 0);
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return set_error(IA3_EHIP, "no HIP device available (%s); libia3 has no CPU fallback",
                     e != hipSuccess ? hipGetErrorString(e) : "0 devices");
  if (device < 0) {
    const char* lr = getenv("LOCAL_RANK");
    device = lr ? atoi(lr) % n : 0;
  }
  if (device >= n) return set_error(IA3_EINVAL, "device %d out of range (%d devices)", device, n);
  IA3_HIP(hipSetDevice(device));
  if (g_stream && g_pid == getpid()) {
    for (size_t i = 0; i < g_live_streams.size(); ++i)
      if (g_live_streams[i] == g_stream) { g_live_streams[i] = g_live_streams.back(); g_live_streams.pop_back(); break; }
    (void)hipStreamDestroy(g_stream); g_stream = nullptr;
  }
  IA3_HIP(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
  g_device = device;
  g_pid = getpid();
  g_live_streams.push_back(g_stream);   // (g_mu is held by the caller, ensure_init)
  t_ctx = ThreadCtx();
  t_ctx.pid = g_pid;
  t_ctx.main = g_stream;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) g_cus = cus;
  return IA3_OK;
}

// the calling thread's context (its own stream unless it is the initialising thread)
static int thread_ctx() {
  if (t_ctx.main && t_ctx.pid == g_pid) return IA3_OK;
  t_ctx = ThreadCtx();
  IA3_HIP(hipSetDevice(g_device));   // the current device is per host thread
  IA3_HIP(hipStreamCreateWithFlags(&t_ctx.main, hipStreamNonBlocking));
  t_ctx.pid = g_pid;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    g_live_streams.push_back(t_ctx.main);
  }
  return IA3_OK;
}

int ensure_init() {
  {
    std::lock_guard<std::mutex> lk(g_mu);
    int rc = do_init(-1);
    if (rc) return rc;
  }
  return thread_ctx();
}
static void ws_flush_deferred();
hipStream_t stream() { return t_ctx.cur ? t_ctx.cur : t_ctx.main; }

// A pinned host block of the calling thread that kernels can write (host pointer + the device's view of it).  Small
// results — the seed count between the detector and the fit, the row table at the end — land here without a copy
// command: a few words written by a kernel are visible to a polling host thread microseconds after the store, where a
// device-to-host copy into pageable memory followed by a stream synchronisation costs 30-50 us.
int host_mailbox(size_t bytes, void** host, void** dev) {
  ThreadCtx& c = t_ctx;
  if (c.mail_bytes < bytes) {
    if (c.mail_host) (void)hipHostFree(c.mail_host);
    c.mail_host = c.mail_dev = nullptr; c.mail_bytes = 0;
    const size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
    if (hipHostMalloc(&c.mail_host, want, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return set_error(IA3_ENOMEM, "pinned mailbox");
    if (hipHostGetDevicePointer(&c.mail_dev, c.mail_host, 0) != hipSuccess) { (void)hipHostFree(c.mail_host); c.mail_host = nullptr; return set_error(IA3_EHIP, "pinned mailbox mapping"); }
    memset(c.mail_host, 0, want);
    c.mail_bytes = want;
  }
  *host = c.mail_host; *dev = c.mail_dev;
  return IA3_OK;
}
void SpinWait::relax() {
  if (++n < 40000) {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#endif
    return;
  }
  if (n < 40400) { sched_yield(); return; }
  timespec ts = {0, 20000};   // 20 us
  nanosleep(&ts, nullptr);
}

// hipStreamSynchronize that spins on an event instead of sleeping (the results it waits for are microseconds away)
int stream_wait_spin(hipStream_t st) {
  ThreadCtx& c = t_ctx;
  if (!c.mail_ev && hipEventCreateWithFlags(&c.mail_ev, hipEventDisableTiming) != hipSuccess) { c.mail_ev = nullptr; IA3_HIP(hipStreamSynchronize(st)); return IA3_OK; }
  IA3_HIP(hipEventRecord(c.mail_ev, st));
  unsigned long long looks = 0;
  for (;;) {
    const hipError_t e = hipEventQuery(c.mail_ev);
    if (e == hipSuccess) return IA3_OK;
    if (e != hipErrorNotReady) return set_error(IA3_EHIP, "stream wait failed: %s", hipGetErrorString(e));
    if (++looks > 2000) { timespec ts = {0, 20000}; nanosleep(&ts, nullptr); }   // a long wait: give the core up between looks
  }
}

// Everything launched while an AuxScope is alive goes to the thread's auxiliary stream, which first waits for the work
// queued on its main stream so far; aux_join() makes the main stream wait for the auxiliary work (typically right
// before the first consumer of its results).
AuxScope::AuxScope() : ok(false) {
  ThreadCtx& c = t_ctx;
  if (!c.main) return;
  if (!c.aux) {
    if (hipStreamCreateWithFlags(&c.aux, hipStreamNonBlocking) != hipSuccess) { c.aux = nullptr; return; }
    if (hipEventCreateWithFlags(&c.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c.join, hipEventDisableTiming) != hipSuccess) return;
  }
  if (hipEventRecord(c.fork, c.main) != hipSuccess || hipStreamWaitEvent(c.aux, c.fork, 0) != hipSuccess) return;
  c.cur = c.aux;
  ok = true;
}
AuxScope::~AuxScope() { t_ctx.cur = nullptr; }
AuxResume::AuxResume() : ok(false) {
  ThreadCtx& c = t_ctx;
  if (!c.main || !c.aux) return;
  c.cur = c.aux;
  ok = true;
}
AuxResume::~AuxResume() { t_ctx.cur = nullptr; }
int aux_join() {
  ThreadCtx& c = t_ctx;
  if (!c.aux) return IA3_OK;
  if (hipEventRecord(c.join, c.aux) != hipSuccess || hipStreamWaitEvent(c.main, c.join, 0) != hipSuccess)
    return set_error(IA3_EHIP, "stream join failed");
  ws_flush_deferred();
  return IA3_OK;
}
void dbg_stamp(const char* what) {
  static const bool on = getenv("IA3_DEBUG_TIMES") != nullptr;
  if (!on) return;
  timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  fprintf(stderr, "[%.1f us] %s\n", ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3, what);
}
int num_cus() { return g_cus; }

// how often the cache went to the driver (ia3_workspace_stats): a steady-state loop should show none
static unsigned long long g_ws_mallocs = 0, g_ws_frees = 0;
static double g_ws_ms = 0.0;
static double ws_now_ms() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

void* ws_get(size_t bytes) {
  if (bytes == 0) bytes = 256;
  std::lock_guard<std::mutex> lk(g_mu);
  const hipStream_t me = stream();
  int best = -1;
  for (size_t i = 0; i < g_ws.size(); ++i)
    if (!g_ws[i].busy && g_ws[i].p && g_ws[i].bytes >= bytes &&
        (best < 0 || g_ws[i].bytes < g_ws[best].bytes))
      best = (int)i;
  if (best >= 0 && g_ws[best].bytes <= 2 * bytes + (1 << 20)) {
    WsEntry& e = g_ws[best];
    // last used on another stream (another host thread, or this thread's other stream): order after that use
    if (e.last && e.last != me && e.ev) (void)hipStreamWaitEvent(me, e.ev, 0);
    e.busy = true;
    return e.p;
  }
  void* p = nullptr;
  const double ta = ws_now_ms();
  if (hipMalloc(&p, bytes) != hipSuccess) {
    for (auto& e : g_ws) if (!e.busy && e.p) { (void)hipFree(e.p); e.p = nullptr; e.bytes = 0; ++g_ws_frees; }
    if (hipMalloc(&p, bytes) != hipSuccess) { set_error(IA3_ENOMEM, "hipMalloc(%zu) failed", bytes); return nullptr; }
  }
  ++g_ws_mallocs; g_ws_ms += ws_now_ms() - ta;
  for (auto& e : g_ws) if (!e.p) { e.p = p; e.bytes = bytes; e.busy = true; e.last = nullptr; return p; }
  g_ws.push_back(WsEntry{p, bytes, true, nullptr, nullptr});
  return p;
}
// idle blocks above this total go back to the driver (largest first): stacks of many different sizes would otherwise
// pile up in a long-running process.  IA3_CACHE_GB overrides the default of three quarters of the device's memory (216 GB
// of 288; it was 64 GB until round 4: the movie pipeline — three movies in correction, twelve images waiting for their
// group fit — parks 40-60 GB of blocks between uses, and every trim is a hipFree that synchronises the device and is
// followed by a hipMalloc of the same size a moment later: a 200 ms stall of all streams in the timeline of
// profiles/r04f; with half the memory as the limit the same stalls came back inside bench.py, whose earlier legs leave
// ~100 GB of blocks of other sizes behind, profiles/r04r).  A hipMalloc that fails frees every idle block and tries again
// (ws_get), so the limit is a courtesy to other processes on the card, not a safety net.
static size_t ws_idle_limit() {
  static const size_t lim = [] {
    const char* e = getenv("IA3_CACHE_GB");
    double gb = e ? atof(e) : 0.0;
    if (!e) {
      size_t fr = 0, tot = 0;
      gb = hipMemGetInfo(&fr, &tot) == hipSuccess ? (double)tot / 1073741824.0 * 0.75 : 64.0;
      if (gb < 16.0) gb = 16.0;
    }
    return (size_t)((gb > 0 ? gb : 0) * 1073741824.0);
  }();
  return lim;
}
static void ws_trim_locked() {
  size_t idle = 0;
  for (auto& e : g_ws) if (!e.busy && e.p) idle += e.bytes;
  while (idle > ws_idle_limit()) {
    int big = -1;
    for (size_t i = 0; i < g_ws.size(); ++i)
      if (!g_ws[i].busy && g_ws[i].p && (big < 0 || g_ws[i].bytes > g_ws[big].bytes)) big = (int)i;
    if (big < 0) break;
    const double ta = ws_now_ms();
    (void)hipFree(g_ws[big].p);   // synchronises the device: rare by construction
    ++g_ws_frees; g_ws_ms += ws_now_ms() - ta;
    idle -= g_ws[big].bytes;
    g_ws[big].p = nullptr; g_ws[big].bytes = 0; g_ws[big].last = nullptr;
  }
}
static void ws_release_locked(void* p, hipStream_t on) {
  for (auto& e : g_ws) if (e.p == p) {
    if (!e.ev && hipEventCreateWithFlags(&e.ev, hipEventDisableTiming) != hipSuccess) e.ev = nullptr;
    if (e.ev) (void)hipEventRecord(e.ev, on);
    e.last = on;
    e.busy = false;
    ws_trim_locked();
    return;
  }
}
// A block returned while an AuxScope is open may still be in use by the auxiliary stream: it stays busy until
// aux_join() has put the join into the main queue.
void ws_put(void* p) {
  if (t_ctx.cur || t_ctx.defer_puts) { t_ctx.deferred.push_back(p); return; }
  std::lock_guard<std::mutex> lk(g_mu);
  ws_release_locked(p, t_ctx.main ? t_ctx.main : g_stream);
}
static void ws_flush_deferred() {
  if (t_ctx.defer_puts) return;   // a PutDefer is open: its owner decides when
  std::lock_guard<std::mutex> lk(g_mu);
  for (void* p : t_ctx.deferred) ws_release_locked(p, t_ctx.main);
  t_ctx.deferred.clear();
}
// Returning a block to the cache records an event on the stream (ws_release_locked): ~2.5 us of host time each, ten
// blocks at the end of the seed stage — 25 us during which the device has nothing to do, because the host has just
// learnt the seed count and has not launched the fit yet.  Inside a PutDefer the blocks are only queued; they go back
// when the owner says so (the fit launches call ws_put_deferred_now) or when the scope ends.
PutDefer::PutDefer() : outer(t_ctx.defer_puts) { t_ctx.defer_puts = true; }
PutDefer::~PutDefer() {
  t_ctx.defer_puts = outer;
  if (!outer && !t_ctx.cur) ws_flush_deferred();
}
void ws_put_deferred_now() {
  if (t_ctx.cur || t_ctx.deferred.empty()) return;
  const bool d = t_ctx.defer_puts;
  t_ctx.defer_puts = false;
  ws_flush_deferred();
  t_ctx.defer_puts = d;
}
// Make sure the cache holds at least `count` blocks a request of `bytes` can take (in use or idle).  A pipeline whose
// depth is bounded knows its peak demand before it starts; finding it out block by block costs a hipMalloc — 70 ms for a
// 1.7 GB block, with every stream of the device stalled — at a moment nobody chose (profiles/r04u).  Best effort: a
// failed allocation ends the reservation quietly.
void ws_reserve(size_t bytes, int count) {
  if (bytes == 0 || count <= 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  int have = 0;
  for (auto& e : g_ws) if (e.p && e.bytes >= bytes && e.bytes <= 2 * bytes + (1 << 20)) ++have;
  for (; have < count; ++have) {
    void* p = nullptr;
    const double ta = ws_now_ms();
    if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return; }
    ++g_ws_mallocs; g_ws_ms += ws_now_ms() - ta;
    bool placed = false;
    for (auto& e : g_ws) if (!e.p) { e.p = p; e.bytes = bytes; e.busy = false; e.last = nullptr; placed = true; break; }
    if (!placed) g_ws.push_back(WsEntry{p, bytes, false, nullptr, nullptr});
  }
}
void ws_release_all() {
  std::lock_guard<std::mutex> lk(g_mu);
  (void)hipDeviceSynchronize();
  for (auto& e : g_ws) if (!e.busy && e.p) { (void)hipFree(e.p); e.p = nullptr; e.bytes = 0; e.last = nullptr; }
}

// ---- profiling ----------------------------------------------------------------------------------
struct ProfRec { const char* name; hipEvent_t a, b; hipStream_t st; };
static bool g_prof = false;
static std::vector<ProfRec> g_recs;

static std::mutex g_prof_mu;
static std::vector<hipEvent_t> g_evpool;   // collected events are reused: creating two per scope cost more than recording them
static hipEvent_t prof_event_locked() {
  if (!g_evpool.empty()) { hipEvent_t e = g_evpool.back(); g_evpool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
ProfScope::ProfScope(const char* name) : slot(-1) {
  if (!g_prof || !stream()) return;
  ProfRec r;
  r.name = name;
  r.st = stream();
  std::lock_guard<std::mutex> lk(g_prof_mu);
  r.a = prof_event_locked();
  r.b = prof_event_locked();
  if (!r.a || !r.b) { if (r.a) g_evpool.push_back(r.a); if (r.b) g_evpool.push_back(r.b); return; }
  (void)hipEventRecord(r.a, r.st);
  g_recs.push_back(r);
  slot = (int)g_recs.size() - 1;
}
ProfScope::~ProfScope() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  (void)hipEventRecord(g_recs[slot].b, g_recs[slot].st);
}

// NumPy's pairwise sum for a contiguous double vector (numpy/_core/src/umath/loops_utils.h.src)
static double np_pairwise(const double* a, int n) {
  if (n < 8) { double r = 0.0; for (int i = 0; i < n; ++i) r += a[i]; return r; }
  if (n <= 128) {
    double r[8];
    for (int k = 0; k < 8; ++k) r[k] = a[k];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) for (int k = 0; k < 8; ++k) r[k] += a[i + k];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise(a, n2) + np_pairwise(a + n2, n - n2);
}

void gaussian_taps(double sigma, double truncate, std::vector<double>& w, int& radius) {
  radius = (int)(truncate * sigma + 0.5);
  double s2 = sigma * sigma;
  w.resize(2 * radius + 1);
  for (int i = -radius; i <= radius; ++i) w[i + radius] = exp(-0.5 / s2 * (double)(i * i));
  double tot = np_pairwise(w.data(), (int)w.size());
  for (auto& v : w) v = v / tot;
}

}  // namespace ia3rt

using namespace ia3rt;

extern "C" {

int ia3_init(int device) {
  std::lock_guard<std::mutex> lk(g_mu);
  return do_init(device);
}
const char* ia3_last_error(void) { return g_err; }
const char* ia3_version(void) { return "ia3-mi355x 0.1.0 (gfx950)"; }
int ia3_device_name(char* buf, int len) {
  int rc = ensure_init();
  if (rc) return rc;
  hipDeviceProp_t prop;
  IA3_HIP(hipGetDeviceProperties(&prop, g_device));
  snprintf(buf, len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return IA3_OK;
}
int ia3_sync(void) {
  int rc = ensure_init();
  if (rc) return rc;
  IA3_HIP(hipStreamSynchronize(stream()));
  return IA3_OK;
}
void* ia3_stream(void) { return ensure_init() ? nullptr : (void*)stream(); }
int ia3_release_workspace(void) { ws_release_all(); return IA3_OK; }
int ia3_workspace_stats(double* out6) {
  if (!out6) return set_error(IA3_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(g_mu);
  double idle = 0, busy = 0, blocks = 0;
  for (auto& e : g_ws) if (e.p) { (e.busy ? busy : idle) += (double)e.bytes; blocks += 1; }
  out6[0] = idle; out6[1] = busy; out6[2] = blocks; out6[3] = (double)g_ws_mallocs; out6[4] = (double)g_ws_frees; out6[5] = g_ws_ms;
  return IA3_OK;
}

int ia3_profile_enable(int on) {
  int rc = ensure_init(); if (rc) return rc;
  g_prof = on != 0;
  return IA3_OK;
}
// Writes "name,count,total_ms\n" lines for everything recorded since the last call, and clears.
int ia3_profile_collect(char* buf, int len) {
  int rc = ensure_init(); if (rc) return rc;
  IA3_HIP(hipDeviceSynchronize());   // events may sit on any thread's stream
  std::lock_guard<std::mutex> plk(g_prof_mu);
  struct Agg { const char* name; int n; double ms; };
  std::vector<Agg> agg;
  for (auto& r : g_recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) ms = 0.f;
    g_evpool.push_back(r.a); g_evpool.push_back(r.b);
    bool found = false;
    for (auto& a : agg) if (strcmp(a.name, r.name) == 0) { a.n++; a.ms += ms; found = true; break; }
    if (!found) agg.push_back({r.name, 1, (double)ms});
  }
  g_recs.clear();
  int off = 0;
  if (buf && len > 0) buf[0] = 0;
  for (auto& a : agg) {
    int w = snprintf(buf + off, off < len ? len - off : 0, "%s,%d,%.6f\n", a.name, a.n, a.ms);
    if (w < 0 || off + w >= len) break;
    off += w;
  }
  return IA3_OK;
}

static int check_shape(int dtype, int Z, int X, int Y) {
  if (dtype != IA3_U16 && dtype != IA3_F32) return set_error(IA3_EINVAL, "unsupported dtype code %d", dtype);
  if (Z <= 0 || X <= 0 || Y <= 0) return set_error(IA3_EINVAL, "bad stack shape (%d,%d,%d)", Z, X, Y);
  return IA3_OK;
}

int ia3_stack_alloc(int dtype, int Z, int X, int Y, ia3_stack** out) {
  int rc = ensure_init(); if (rc) return rc;
  rc = check_shape(dtype, Z, X, Y); if (rc) return rc;
  size_t bytes = (size_t)Z * X * Y * esize(dtype);
  // stacks come from the same stream-ordered cache as the scratch buffers: a chain of operators allocates and frees
  // dozens of 0.4 GB stacks per movie, and hipMalloc / hipFree (which synchronises the device) cost 1-2 ms each.
  // A freed block is handed out again only after the work queued on the freeing thread's stream (ws_get waits for
  // that event when another stream takes it over): free a stack from the thread that used it last.
  void* d = ws_get(bytes);
  if (!d) return IA3_ENOMEM;
  *out = new ia3_stack{d, dtype, Z, X, Y, true, bytes, (void*)t_ctx.main};
  return IA3_OK;
}
// Host array -> resident stack.  Measured on MI355X (profiles/r02a/probe.log, 839 MB float32 / 419 MB uint16 stacks): one
// hipMemcpyAsync from pageable memory runs at 56 GB/s (the runtime pins the caller's pages piecewise and copies from
// them directly), which is the PCIe rate of a pinned buffer; cutting the array into STAGE_BYTES pieces that helper
// threads copy into a ring of pinned staging buffers while the calling thread queues one asynchronous copy per piece
// reaches 35 / 51 / 51 GB/s with 2 / 4 / 8 helpers — the extra host-side copy costs more than it hides.  The plain
// copy is therefore the default; IA3_UPLOAD_THREADS / IA3_TUNE_UPLOAD_THREADS > 0 selects the staged form (useful where
// pinning on the fly is slow, e.g. memory-mapped files).  Overlap with compute comes from running several images at
// once (ia3_fit_fovs), not from this call, which returns when the stack is resident.
namespace {
constexpr size_t STAGE_BYTES = 16u << 20;
constexpr int MAX_STAGE = 8;
struct Staging {
  void* buf[MAX_STAGE] = {};
  hipEvent_t ev[MAX_STAGE] = {};
  int n = 0;
  pid_t pid = 0;
  ~Staging() {   // a host thread that ends gives its pinned ring back (16 MB a buffer); not ours after a fork
    if (pid != getpid()) return;
    for (int i = 0; i < MAX_STAGE; ++i) {
      if (ev[i]) { (void)hipEventSynchronize(ev[i]); (void)hipEventDestroy(ev[i]); }
      if (buf[i]) (void)hipHostFree(buf[i]);
    }
  }
};
thread_local Staging t_stage;
std::atomic<int> g_upload_threads{-1};   // -1: from IA3_UPLOAD_THREADS (default 0)
int upload_threads() {
  int n = g_upload_threads.load(std::memory_order_relaxed);
  if (n < 0) {
    const char* e = getenv("IA3_UPLOAD_THREADS");
    n = e ? atoi(e) : 0;
    n = n < 0 ? 0 : (n > MAX_STAGE ? MAX_STAGE : n);
    g_upload_threads.store(n, std::memory_order_relaxed);
  }
  return n;
}
int staging_ready(int want) {
  Staging& s = t_stage;
  if (s.pid != getpid()) { s = Staging(); s.pid = getpid(); }
  for (; s.n < want; ++s.n) {
    if (hipHostMalloc(&s.buf[s.n], STAGE_BYTES, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&s.ev[s.n], hipEventDisableTiming) != hipSuccess)
      return set_error(IA3_ENOMEM, "cannot allocate pinned staging buffers");
  }
  return IA3_OK;
}
// dst (device) <- src (pageable host), queued on st; returns after the last piece has been queued AND copied out of
// the staging ring (the caller synchronises the stream)
int staged_h2d(void* dst, const void* src, size_t bytes, hipStream_t st) {
  const int T = upload_threads();
  if (T == 0 || bytes < 2 * STAGE_BYTES) {
    IA3_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
    return IA3_OK;
  }
  int rc = staging_ready(T); if (rc) return rc;
  Staging& sg = t_stage;
  const size_t pieces = (bytes + STAGE_BYTES - 1) / STAGE_BYTES;
  // filled[k] = piece k sits in buffer k % T; queued[b] = number of pieces of buffer b whose H2D copy has been queued
  std::vector<std::atomic<int>> filled(pieces);
  for (auto& f : filled) f.store(0, std::memory_order_relaxed);
  std::atomic<long long> queued[MAX_STAGE];
  for (int b = 0; b < T; ++b) queued[b].store(0, std::memory_order_relaxed);
  std::atomic<int> failed{0};
  const int dev = g_device;
  auto helper = [&](int b) {
    (void)hipSetDevice(dev);
    long long mine = 0;   // pieces this helper has filled so far
    for (size_t k = (size_t)b; k < pieces; k += (size_t)T, ++mine) {
      if (mine > 0) {   // the copy that read this buffer last must have been queued, then finished
        while (queued[b].load(std::memory_order_acquire) < mine && !failed.load()) sched_yield();
        if (failed.load() || hipEventSynchronize(sg.ev[b]) != hipSuccess) { failed.store(1); return; }
      }
      const size_t off = k * STAGE_BYTES, n = bytes - off < STAGE_BYTES ? bytes - off : STAGE_BYTES;
      memcpy(sg.buf[b], (const char*)src + off, n);
      filled[k].store(1, std::memory_order_release);
    }
  };
  std::vector<std::thread> th;
  for (int b = 0; b < T; ++b) th.emplace_back(helper, b);
  hipError_t e = hipSuccess;
  for (size_t k = 0; k < pieces && e == hipSuccess; ++k) {
    const int b = (int)(k % (size_t)T);
    while (!filled[k].load(std::memory_order_acquire) && !failed.load()) sched_yield();
    if (failed.load()) break;
    const size_t off = k * STAGE_BYTES, n = bytes - off < STAGE_BYTES ? bytes - off : STAGE_BYTES;
    e = hipMemcpyAsync((char*)dst + off, sg.buf[b], n, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipEventRecord(sg.ev[b], st);
    if (e != hipSuccess) failed.store(1);
    queued[b].fetch_add(1, std::memory_order_release);
  }
  if (e != hipSuccess) failed.store(1);
  for (auto& t : th) t.join();
  if (failed.load()) return set_error(IA3_EHIP, "staged H2D copy failed: %s", hipGetErrorString(e));
  return IA3_OK;
}
}  // namespace
extern "C++" {
namespace ia3rt {
int set_upload_threads(int n) {
  if (n < 0 || n > MAX_STAGE) return set_error(IA3_EINVAL, "IA3_TUNE_UPLOAD_THREADS: 0..%d", MAX_STAGE);
  g_upload_threads.store(n, std::memory_order_relaxed);
  return IA3_OK;
}
}  // namespace ia3rt
}  // extern "C++"

int ia3_stack_upload(const void* host, int dtype, int Z, int X, int Y, ia3_stack** out) {
  if (!host) return set_error(IA3_EINVAL, "null host pointer");
  int rc = ia3_stack_alloc(dtype, Z, X, Y, out); if (rc) return rc;
  rc = staged_h2d((*out)->d, host, (*out)->bytes, stream());
  hipError_t e = hipSuccess;
  if (!rc) e = hipStreamSynchronize(stream());
  if (rc || e != hipSuccess) {
    ia3_stack_free(*out); *out = nullptr;
    return rc ? rc : set_error(IA3_EHIP, "H2D copy failed: %s", hipGetErrorString(e));
  }
  return IA3_OK;
}
// Raw movie file -> resident uint16 stack (what DaxReader.loadAll + an upload do, visual_tools.py:974-1083): the file
// is read in STAGE_BYTES pieces into two of the calling thread's pinned staging buffers and every piece is sent with an
// async copy while the next one is being read, so a movie costs max(file read, PCIe) instead of read + pageable copy.
namespace {
__global__ void bswap16_k(uint16_t* p, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { uint16_t v = p[i]; p[i] = (uint16_t)((v >> 8) | (v << 8)); }
}
}  // namespace

int ia3_stack_load_file(const char* path, long long offset_bytes, int frames, int X, int Y, int big_endian,
                        ia3_stack** out) {
  int rc = ensure_init(); if (rc) return rc;
  if (!path || !out || frames < 1 || X < 1 || Y < 1 || offset_bytes < 0) return set_error(IA3_EINVAL, "bad load arguments");
  int fd = open(path, O_RDONLY);
  if (fd < 0) return set_error(IA3_EINVAL, "cannot open %s", path);
  struct stat stt;
  const size_t total = (size_t)frames * X * Y * 2;
  if (fstat(fd, &stt) != 0 || (size_t)stt.st_size < (size_t)offset_bytes + total) {
    close(fd);
    return set_error(IA3_EINVAL, "%s holds fewer than %d frames of %d x %d uint16", path, frames, X, Y);
  }
  rc = staging_ready(2);
  if (!rc) rc = ia3_stack_alloc(IA3_U16, frames, X, Y, out);
  if (rc) { close(fd); return rc; }
  Staging& sg = t_stage;
  hipStream_t st = stream();
  hipError_t e = hipSuccess;
  size_t done = 0;
  bool used[2] = {false, false};
  for (int k = 0; done < total && e == hipSuccess; ++k) {
    const int b = k & 1;
    if (used[b]) e = hipEventSynchronize(sg.ev[b]);   // the copy that last read this buffer has finished
    if (e != hipSuccess) break;
    const size_t want = total - done < STAGE_BYTES ? total - done : STAGE_BYTES;
    size_t got = 0;
    while (got < want) {
      ssize_t r = pread(fd, (char*)sg.buf[b] + got, want - got, (off_t)(offset_bytes + done + got));
      if (r <= 0) { close(fd); ia3_stack_free(*out); *out = nullptr; return set_error(IA3_EINVAL, "short read from %s", path); }
      got += (size_t)r;
    }
    e = hipMemcpyAsync((char*)(*out)->d + done, sg.buf[b], want, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipEventRecord(sg.ev[b], st);
    used[b] = true;
    done += want;
  }
  close(fd);
  if (e == hipSuccess && big_endian) {
    const size_t n = total / 2;
    hipLaunchKernelGGL(bswap16_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (uint16_t*)(*out)->d, n);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) {
    ia3_stack_free(*out); *out = nullptr;
    return set_error(IA3_EHIP, "movie upload failed: %s", hipGetErrorString(e));
  }
  return IA3_OK;
}
// io_tools/load.py:524-550 split_im_by_channels on a resident raw movie: frames start, start+step, ... (Z of them)
// every step-th frame of a movie into a stack of its own: 16 bytes per thread and access (the driver's rectangle copy moved a
// 419 MB channel in 0.49 ms, four of them per movie)
__global__ __launch_bounds__(256) void frame_gather_k(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t plane16,
                                                      size_t src_stride16) {
  const uint4* s = src + (size_t)blockIdx.y * src_stride16;
  uint4* d = dst + (size_t)blockIdx.y * plane16;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < plane16; i += (size_t)gridDim.x * 256) d[i] = s[i];
}
int ia3_stack_deinterleave(const ia3_stack* raw, int start, int step, int Z, ia3_stack** out) {
  int rc = ensure_init(); if (rc) return rc;
  if (!raw || !out) return set_error(IA3_EINVAL, "null argument");
  if (start < 0 || step < 1 || Z < 1 || (long long)start + (long long)(Z - 1) * step >= raw->Z)
    return set_error(IA3_EINVAL, "frames %d + k*%d (k < %d) fall outside a movie of %d frames", start, step, Z, raw->Z);
  rc = ia3_stack_alloc(raw->dtype, Z, raw->X, raw->Y, out); if (rc) return rc;
  const size_t pb = (size_t)raw->X * raw->Y * esize(raw->dtype);
  hipError_t e;
  const char* first = (const char*)raw->d + (size_t)start * pb;
  if (pb % 16 == 0 && ((uintptr_t)first & 15) == 0 && ((uintptr_t)(*out)->d & 15) == 0 && Z <= 65535) {
    const size_t plane16 = pb / 16;
    unsigned bx = (unsigned)((plane16 + 255) / 256);
    if (bx > 512) bx = 512;
    hipLaunchKernelGGL(frame_gather_k, dim3(bx, (unsigned)Z), dim3(256), 0, stream(), (const uint4*)first, (uint4*)(*out)->d, plane16,
                       plane16 * (size_t)step);
    e = hipGetLastError();
  } else {
    e = hipMemcpy2DAsync((*out)->d, pb, first, pb * step, pb, Z, hipMemcpyDeviceToDevice, stream());
  }
  if (e != hipSuccess) {
    ia3_stack_free(*out); *out = nullptr;
    return set_error(IA3_EHIP, "frame gather failed: %s", hipGetErrorString(e));
  }
  return IA3_OK;
}
// Buffers made by ia3_buffer_upload hold data that stays constant over a run (illumination / bleedthrough / chromatic
// profiles): a consumer may keep a small summary of one beside it — the cubic warp keeps the range of a displacement
// field's z component (warp.hip), 0.2 ms to compute, needed by every warp with that field.  One slot of four floats per
// buffer, made on first request; state 0 = empty, 1 = being filled by some thread, 2 = filled and visible to every stream.
}  // extern "C"
namespace ia3rt {
// slot of `buf` if it is such a buffer, else nullptr.  *fill = true: the caller computes the summary into the slot on its
// stream, synchronises that stream and calls const_note_filled; *ready = true: the slot can be read by any stream.
float* const_note(const void* buf, bool* ready, bool* fill) {
  *ready = *fill = false;
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_const_bufs.find((void*)buf);
  if (it == g_const_bufs.end()) return nullptr;
  ConstNote& n = it->second;
  if (n.state == 2) { *ready = true; return n.slot; }
  if (n.state == 1) return nullptr;                        // another thread is at it: this call works without the note
  if (!n.slot && hipMalloc((void**)&n.slot, 4 * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); n.slot = nullptr; return nullptr; }
  n.state = 1;
  *fill = true;
  return n.slot;
}
void const_note_filled(const void* buf, bool ok) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_const_bufs.find((void*)buf);
  if (it != g_const_bufs.end()) it->second.state = ok ? 2 : 0;
}
}  // namespace ia3rt
extern "C" {
int ia3_buffer_upload(const void* host, size_t bytes, void** devptr) {
  int rc = ensure_init(); if (rc) return rc;
  if (!host || !devptr || bytes == 0) return set_error(IA3_EINVAL, "bad buffer arguments");
  void* d = nullptr;
  if (hipMalloc(&d, bytes) != hipSuccess) return set_error(IA3_ENOMEM, "hipMalloc(%zu) failed", bytes);
  hipError_t e = hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, stream());
  if (e == hipSuccess) e = hipStreamSynchronize(stream());
  if (e != hipSuccess) { (void)hipFree(d); return set_error(IA3_EHIP, "H2D copy failed: %s", hipGetErrorString(e)); }
  *devptr = d;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    g_const_bufs[d] = ConstNote();
  }
  return IA3_OK;
}
void ia3_buffer_free(void* devptr) {
  if (devptr && g_pid == getpid()) {
    (void)hipStreamSynchronize(stream());
    float* slot = nullptr;
    {
      std::lock_guard<std::mutex> lk(g_mu);
      auto it = g_const_bufs.find(devptr);
      if (it != g_const_bufs.end()) { slot = it->second.slot; g_const_bufs.erase(it); }
    }
    if (slot) (void)hipFree(slot);
    (void)hipFree(devptr);
  }
}
int ia3_stack_wrap(void* devptr, int dtype, int Z, int X, int Y, ia3_stack** out) {
  int rc = ensure_init(); if (rc) return rc;
  rc = check_shape(dtype, Z, X, Y); if (rc) return rc;
  if (!devptr) return set_error(IA3_EINVAL, "null device pointer");
  *out = new ia3_stack{devptr, dtype, Z, X, Y, false, (size_t)Z * X * Y * esize(dtype)};
  return IA3_OK;
}
int ia3_stack_download(const ia3_stack* s, void* host) {
  if (!s || !host) return set_error(IA3_EINVAL, "null argument");
  // what produces the stack first, the copy afterwards: a copy queued behind kernels waits for them on a copy engine's
  // ring, and with it every other host thread's device-to-host copy that shares the ring
  IA3_HIP(hipStreamSynchronize(stream()));
  IA3_HIP(hipMemcpyAsync(host, s->d, s->bytes, hipMemcpyDeviceToHost, stream()));
  IA3_HIP(hipStreamSynchronize(stream()));
  return IA3_OK;
}
int ia3_stack_info(const ia3_stack* s, int* dtype, int* Z, int* X, int* Y, void** devptr) {
  if (!s) return set_error(IA3_EINVAL, "null stack");
  if (dtype) *dtype = s->dtype;
  if (Z) *Z = s->Z;
  if (X) *X = s->X;
  if (Y) *Y = s->Y;
  if (devptr) *devptr = s->d;
  return IA3_OK;
}
void ia3_stack_free(ia3_stack* s) {
  if (!s) return;
  if (s->owned && s->d) {
    // freed by another thread than the one that allocated it (a finaliser, a hand-over between workers): the block
    // must not be reused before the allocating thread's queued work either, so this thread's stream waits for it
    hipStream_t home = (hipStream_t)s->home, me = t_ctx.main ? t_ctx.main : g_stream;
    if (home && me && home != me && g_pid == getpid()) {
      static thread_local hipEvent_t ev = nullptr;
      if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
      // (a thread that has ended drained its stream on the way out and is no longer in the list)
      std::lock_guard<std::mutex> lk(g_mu);
      if (ev && stream_alive_locked(home) && hipEventRecord(ev, home) == hipSuccess) (void)hipStreamWaitEvent(me, ev, 0);
    }
    ws_put(s->d);
  }
  delete s;
}

}  // extern "C"
