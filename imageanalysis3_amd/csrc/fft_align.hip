// Drift estimation on the device (rocFFT through the hipFFT API, float64).
//
//  (a10) alignment_tools.py:286-353 fftalign_2d / fft3d_from2d — integer shift from max-projections:
//        normalised full cross-correlation (fftconvolve with the flipped target), argmax inside a
//        ±max_disp window.  z-max projection = one HBM pass over each stack.
//  (a9)  skimage.registration.phase_cross_correlation as called at correction_tools/alignment.py:631 —
//        published algorithm (Guizar-Sicairos et al. 2008): cross-power spectrum, integer peak, then a
//        matrix-multiply upsampled DFT in a ceil(1.5·u)³ window around it.  Pinned against scikit-image 0.18.3
//        (tests/golden/phase.npz) for the un-normalised correlation.
//
// All FFT-side arithmetic is float64 (complex128), the precision SciPy/skimage use for uint16 input.
#include "ia3_rt.h"
#include <hipfft/hipfft.h>
#include <unistd.h>
#include <vector>
#include <mutex>
#include <sys/syscall.h>
#include <math.h>

using namespace ia3rt;

namespace {

typedef hipfftDoubleComplex cplx;

int next_fast_len(int n) {  // smallest 2^a 3^b 5^c 7^d >= n
  for (int m = n;; ++m) {
    int k = m;
    for (int p : {2, 3, 5, 7}) while (k % p == 0) k /= p;
    if (k == 1) return m;
  }
}

#define IA3_FFT(expr)                                                                          \
  do {                                                                                         \
    hipfftResult _r = (expr);                                                                  \
    if (_r != HIPFFT_SUCCESS) return set_error(IA3_EHIP, "%s failed: hipfft error %d", #expr, (int)_r); \
  } while (0)

// rocFFT plans cost tens of milliseconds to build (kernel selection, twiddle tables, work buffer); drift alignment
// runs the same few transforms for every crop of every image, so plans are kept: ONE pool for the process, keyed by
// (type, dims).  A plan with its work buffer must not run on two streams at once, so a thread LEASES a plan for the
// duration of a call (PlanLease, released after the call's final synchronisation) and another thread that needs the same
// transform meanwhile builds a second one.  Plans are never destroyed when a thread ends — round 1 kept a cache per thread
// and released it from the thread's destructor, and a plan destroyed there while other threads were creating plans or
// launching crashed inside the runtime (std::map erase under rocfft_plan_destroy; scripts/stress_threads.py) — only the
// least recently used idle plan goes when the pool is full, under the pool's lock.  What is left at process exit is left
// to process teardown (the HIP runtime may already be gone by then).
struct PlanEntry { int type, n0, n1, n2; hipfftHandle h; unsigned long long used; bool busy; };
constexpr size_t MAX_PLANS = 48;
struct PlanPool {
  std::vector<PlanEntry> plans;
  pid_t pid = 0;
  unsigned long long clock = 0;
  std::mutex mu;   // guards the table AND serialises plan construction / destruction
};
static PlanPool& plan_pool() { static PlanPool* p = new PlanPool(); return *p; }   // never destructed

struct PlanLease {
  hipfftHandle h = 0;
  bool held = false;
  PlanLease() = default;
  PlanLease(const PlanLease&) = delete;
  PlanLease& operator=(const PlanLease&) = delete;
  ~PlanLease() { release(); }
  void release() {
    if (!held) return;
    PlanPool& c = plan_pool();
    std::lock_guard<std::mutex> lk(c.mu);
    if (c.pid == getpid())
      for (auto& e : c.plans) if (e.h == h && e.busy) { e.busy = false; break; }
    held = false;
  }
};

static int get_plan(int type, int n0, int n1, int n2, hipStream_t st, PlanLease& out) {
  PlanPool& c = plan_pool();
  std::lock_guard<std::mutex> lk(c.mu);
  if (c.pid != getpid()) { c.plans.clear(); c.pid = getpid(); }   // handles do not survive fork()
  for (auto& e : c.plans)
    if (!e.busy && e.type == type && e.n0 == n0 && e.n1 == n1 && e.n2 == n2) {
      if (hipfftSetStream(e.h, st) != HIPFFT_SUCCESS) return set_error(IA3_EHIP, "hipfftSetStream failed");
      e.used = ++c.clock; e.busy = true;
      out.h = e.h; out.held = true;
      return IA3_OK;
    }
  if (c.plans.size() >= MAX_PLANS) {   // make room: the least recently used idle plan (its last call has synchronised)
    int victim = -1;
    for (size_t i = 0; i < c.plans.size(); ++i)
      if (!c.plans[i].busy && (victim < 0 || c.plans[i].used < c.plans[victim].used)) victim = (int)i;
    if (victim >= 0) { hipfftDestroy(c.plans[victim].h); c.plans.erase(c.plans.begin() + victim); }
  }
  hipfftHandle h;
  hipfftResult r = n2 > 0 ? hipfftPlan3d(&h, n0, n1, n2, (hipfftType)type) : hipfftPlan2d(&h, n0, n1, (hipfftType)type);
  if (r != HIPFFT_SUCCESS) return set_error(IA3_EHIP, "hipfft plan (%d x %d x %d) failed: error %d", n0, n1, n2, (int)r);
  r = hipfftSetStream(h, st);
  if (r != HIPFFT_SUCCESS) { hipfftDestroy(h); return set_error(IA3_EHIP, "hipfftSetStream failed: error %d", (int)r); }
  c.plans.push_back(PlanEntry{type, n0, n1, n2, h, ++c.clock, true});
  out.h = h; out.held = true;
  return IA3_OK;
}

template <class T> __device__ __forceinline__ double ldv(const T* p, size_t i) { return (double)p[i]; }

// out[x,y] = max_z im[z, x0+x, y0+y]   (sub-box x0..x0+nx, y0..y0+ny)
template <class T>
__global__ void maxproj_z_k(const T* __restrict__ im, int Z, int X, int Y, int x0, int nx, int y0, int ny,
                            double* __restrict__ out) {
  int y = blockIdx.x * 64 + (threadIdx.x & 63), x = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= nx || y >= ny) return;
  double m = -INFINITY;
  for (int z = 0; z < Z; ++z) { double v = ldv(im, ((size_t)z * X + x0 + x) * Y + y0 + y); m = v > m ? v : m; }
  out[(size_t)x * ny + y] = m;
}
// out[z,x] = max_y im[z, x0+x, y0..y0+ny)
template <class T>
__global__ void maxproj_y_k(const T* __restrict__ im, int Z, int X, int Y, int x0, int nx, int y0, int ny,
                            double* __restrict__ out) {
  const int row = blockIdx.x;  // z * nx + x
  const int z = row / nx, x = row % nx;
  const T* p = im + ((size_t)z * X + x0 + x) * Y + y0;
  double m = -INFINITY;
  for (int y = threadIdx.x; y < ny; y += 64) { double v = (double)p[y]; m = v > m ? v : m; }
  for (int s = 1; s < 64; s <<= 1) { double o = __shfl_xor(m, s); m = o > m ? o : m; }
  if (threadIdx.x == 0) out[row] = m;
}

// single-block mean / std (np.mean, np.std ddof 0) of n doubles -> stats[0]=mean, stats[1]=std
__global__ __launch_bounds__(1024) void mean_std_k(const double* __restrict__ a, size_t n, double* stats) {
  __shared__ double sh[1024];
  double s = 0;
  for (size_t i = threadIdx.x; i < n; i += 1024) s += a[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int k = 512; k > 0; k >>= 1) { if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k]; __syncthreads(); }
  const double mean = sh[0] / (double)n;
  __syncthreads();
  double q = 0;
  for (size_t i = threadIdx.x; i < n; i += 1024) { double d = a[i] - mean; q += d * d; }
  sh[threadIdx.x] = q;
  __syncthreads();
  for (int k = 512; k > 0; k >>= 1) { if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k]; __syncthreads(); }
  if (threadIdx.x == 0) { stats[0] = mean; stats[1] = sqrt(sh[0] / (double)n); }
}

// dst (Fx x Fy, zero padded) = (src - mean)/std, optionally flipped in both axes (im2[::-1, ::-1])
__global__ void pad_norm_k(const double* __restrict__ src, int sx, int sy, const double* __restrict__ stats,
                           int flip, double* __restrict__ dst, int Fx, int Fy) {
  int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (j >= Fy) return;
  double v = 0.0;
  if (i < sx && j < sy) {
    int si = flip ? sx - 1 - i : i, sj = flip ? sy - 1 - j : j;
    v = (src[(size_t)si * sy + sj] - stats[0]) / stats[1];
  }
  dst[(size_t)i * Fy + j] = v;
}
__global__ void cmul_k(cplx* __restrict__ a, const cplx* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  cplx x = a[i], y = b[i];
  a[i] = cplx{x.x * y.x - x.y * y.y, x.x * y.y + x.y * y.x};
}
// argmax over cor[0:cx, 0:cy] (leading dim Fy) of (inside window ? cor*scale : 0); first index wins ties
__global__ __launch_bounds__(1024) void window_argmax_k(const double* __restrict__ cor, int cx, int cy, int Fy,
                                                       double scale, int x_min, int x_max, int y_min, int y_max,
                                                       long long* __restrict__ out_idx, double* __restrict__ out_val) {
  __shared__ double sv[1024];
  __shared__ long long si[1024];
  double bv = -INFINITY; long long bi = 0x7fffffffffffffffLL;
  const long long n = (long long)cx * cy;
  for (long long k = threadIdx.x; k < n; k += 1024) {
    int i = (int)(k / cy), j = (int)(k % cy);
    bool in = i >= x_min && i < x_max && j >= y_min && j < y_max;
    double v = in ? cor[(size_t)i * Fy + j] * scale : 0.0;
    if (v > bv) { bv = v; bi = k; }   // k increases per thread -> first occurrence kept
  }
  sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      double ov = sv[threadIdx.x + s]; long long oi = si[threadIdx.x + s];
      if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { *out_idx = si[0]; *out_val = sv[0]; }
}

// fftalign_2d on device-resident 2-D float64 images (alignment_tools.py:286-328)
int fftalign2d_dev(const double* im1, int s1x, int s1y, const double* im2, int s2x, int s2y,
                   const double center[2], double max_disp, int out[2]) {
  hipStream_t st = stream();
  if (s1x < 1 || s1y < 1 || s2x < 1 || s2y < 1) return set_error(IA3_EINVAL, "empty image in fftalign_2d");
  const int cx = s1x + s2x - 1, cy = s1y + s2y - 1;
  const int Fx = next_fast_len(cx), Fy = next_fast_len(cy);
  const size_t nreal = (size_t)Fx * Fy, ncplx = (size_t)Fx * (Fy / 2 + 1);
  Scratch a(nreal * sizeof(double)), b(nreal * sizeof(double)), fa(ncplx * sizeof(cplx)), fb(ncplx * sizeof(cplx)),
      stats(4 * sizeof(double)), res(sizeof(long long) + sizeof(double));
  if (!a.p || !b.p || !fa.p || !fb.p || !stats.p || !res.p) return IA3_ENOMEM;
  ProfScope ps("fftalign_2d");
  hipLaunchKernelGGL(mean_std_k, dim3(1), dim3(1024), 0, st, im1, (size_t)s1x * s1y, stats.as<double>());
  hipLaunchKernelGGL(mean_std_k, dim3(1), dim3(1024), 0, st, im2, (size_t)s2x * s2y, stats.as<double>() + 2);
  dim3 g((Fy + 255) / 256, Fx);
  hipLaunchKernelGGL(pad_norm_k, g, dim3(256), 0, st, im1, s1x, s1y, (const double*)stats.as<double>(), 0, a.as<double>(), Fx, Fy);
  hipLaunchKernelGGL(pad_norm_k, g, dim3(256), 0, st, im2, s2x, s2y, (const double*)(stats.as<double>() + 2), 1, b.as<double>(), Fx, Fy);
  PlanLease fwd, inv;   // held until this call has synchronised (end of scope)
  { int prc = get_plan(HIPFFT_D2Z, Fx, Fy, 0, st, fwd); if (prc) return prc; }
  { int prc = get_plan(HIPFFT_Z2D, Fx, Fy, 0, st, inv); if (prc) return prc; }
  IA3_FFT(hipfftExecD2Z(fwd.h, a.as<double>(), fa.as<cplx>()));
  IA3_FFT(hipfftExecD2Z(fwd.h, b.as<double>(), fb.as<cplx>()));
  hipLaunchKernelGGL(cmul_k, dim3((unsigned)((ncplx + 255) / 256)), dim3(256), 0, st, fa.as<cplx>(), (const cplx*)fb.as<cplx>(), ncplx);
  IA3_FFT(hipfftExecZ2D(inv.h, fa.as<cplx>(), a.as<double>()));
  // window (alignment_tools.py:301-308)
  const double c0 = center[0] + cx / 2.0, c1 = center[1] + cy / 2.0;
  auto clampi = [](double v, int hi) { v = v < 0 ? 0 : v; v = v > hi ? hi : v; return (int)v; };
  const int x_min = clampi(c0 - max_disp, cx), x_max = clampi(c0 + max_disp, cx);
  const int y_min = clampi(c1 - max_disp, cy), y_max = clampi(c1 + max_disp, cy);
  long long* d_idx = res.as<long long>();
  double* d_val = (double*)(d_idx + 1);
  hipLaunchKernelGGL(window_argmax_k, dim3(1), dim3(1024), 0, st, (const double*)a.as<double>(), cx, cy, Fy,
                     1.0 / (double)nreal, x_min, x_max, y_min, y_max, d_idx, d_val);
  IA3_KCHECK();
  long long idx = 0;
  IA3_HIP(hipMemcpyAsync(&idx, d_idx, sizeof(idx), hipMemcpyDeviceToHost, st));
  IA3_HIP(hipStreamSynchronize(st));
  const int yy = (int)(idx / cy), xx = (int)(idx % cy);
  out[0] = -(cx / 2) + yy;   // -floor(shape/2) + [y, x]   (:327)
  out[1] = -(cy / 2) + xx;
  return IA3_OK;
}

template <class T>
void launch_maxproj_z(const ia3_stack* s, int x0, int nx, int y0, int ny, double* out, hipStream_t st) {
  dim3 g((ny + 63) / 64, (nx + 3) / 4);
  hipLaunchKernelGGL((maxproj_z_k<T>), g, dim3(256), 0, st, (const T*)s->d, s->Z, s->X, s->Y, x0, nx, y0, ny, out);
}
template <class T>
void launch_maxproj_y(const ia3_stack* s, int x0, int nx, int y0, int ny, double* out, hipStream_t st) {
  hipLaunchKernelGGL((maxproj_y_k<T>), dim3((unsigned)(s->Z * nx)), dim3(64), 0, st, (const T*)s->d, s->Z, s->X, s->Y, x0, nx, y0, ny, out);
}

// ---- phase cross-correlation -----------------------------------------------------------------------
template <class T>
__global__ void to_cplx_k(const T* __restrict__ a, cplx* __restrict__ o, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) o[i] = cplx{(double)a[i], 0.0};
}
// prod = A * conj(B); optional phase normalisation prod /= max(|prod|, 100 eps)
__global__ void cross_power_k(cplx* __restrict__ a, const cplx* __restrict__ b, size_t n, int phase_norm) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  cplx x = a[i], y = b[i];
  double re = x.x * y.x + x.y * y.y, im = x.y * y.x - x.x * y.y;
  if (phase_norm) {
    double m = hypot(re, im);
    const double lim = 100.0 * 2.220446049250313e-16;
    m = m > lim ? m : lim;
    re /= m; im /= m;
  }
  a[i] = cplx{re, im};
}
__global__ void conj_k(cplx* __restrict__ a, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i].y = -a[i].y;
}
// multi-block |.|² argmax: per-block best -> partial arrays
__global__ __launch_bounds__(256) void abs_argmax_part_k(const cplx* __restrict__ a, size_t n, double* pv, long long* pi) {
  __shared__ double sv[256];
  __shared__ long long si[256];
  double bv = -1.0; long long bi = 0x7fffffffffffffffLL;
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) {
    double v = a[k].x * a[k].x + a[k].y * a[k].y;
    if (v > bv) { bv = v; bi = (long long)k; }
  }
  sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      double ov = sv[threadIdx.x + s]; long long oi = si[threadIdx.x + s];
      if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { pv[blockIdx.x] = sv[0]; pi[blockIdx.x] = si[0]; }
}
// sum of |a|^2: fixed grid of partial sums, then one block adds the partials in index order (deterministic)
constexpr int ABS2_BLOCKS = 1024;
__global__ __launch_bounds__(256) void abs2_part_k(const cplx* __restrict__ a, size_t n, double* __restrict__ part) {
  __shared__ double sh[256];
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)ABS2_BLOCKS * 256) s += a[i].x * a[i].x + a[i].y * a[i].y;
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) { if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k]; __syncthreads(); }
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(1024) void abs2_sum_k(const double* __restrict__ part, double* out) {
  __shared__ double sh[1024];
  sh[threadIdx.x] = threadIdx.x < ABS2_BLOCKS ? part[threadIdx.x] : 0.0;
  __syncthreads();
  for (int k = 512; k > 0; k >>= 1) { if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k]; __syncthreads(); }
  if (threadIdx.x == 0) *out = sh[0];
}
// DFT kernel matrix K[r, n] = exp(-2πi (r - off) * fftfreq(N, u)[n]),  r < R, n < N
__global__ void dft_kernel_k(cplx* __restrict__ K, int R, int N, double off, double u) {
  int n = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (n >= N) return;
  int kf = n < (N + 1) / 2 ? n : n - N;
  double turns = ((double)r - off) * ((double)kf / ((double)N * u));
  double s, c;
  sincospi(-2.0 * turns, &s, &c);
  K[(size_t)r * N + n] = cplx{c, s};
}
// out[r, m] = sum_n K[r, n] * in[m, n]     (np.tensordot(K, data, axes=(1, -1)) with data flattened to (M, N))
// tile: 16 r x 64 m per 256-thread block, 4 m per thread, n staged through LDS in chunks of 32
__global__ __launch_bounds__(256) void dft_contract_k(const cplx* __restrict__ K, const cplx* __restrict__ in,
                                                      cplx* __restrict__ out, int R, int M, int N, int ldk, int ldi) {
  __shared__ cplx sk[16][33];
  __shared__ cplx sd[64][33];
  const int tr = threadIdx.x >> 4, tm = threadIdx.x & 15;
  const int r0 = blockIdx.y * 16, m0 = blockIdx.x * 64;
  cplx acc[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
  for (int n0 = 0; n0 < N; n0 += 32) {
    for (int e = threadIdx.x; e < 16 * 32; e += 256) {
      int rr = e >> 5, nn = e & 31;
      sk[rr][nn] = (r0 + rr < R && n0 + nn < N) ? K[(size_t)(r0 + rr) * ldk + n0 + nn] : cplx{0, 0};
    }
    for (int e = threadIdx.x; e < 64 * 32; e += 256) {
      int mm = e >> 5, nn = e & 31;
      sd[mm][nn] = (m0 + mm < M && n0 + nn < N) ? in[(size_t)(m0 + mm) * ldi + n0 + nn] : cplx{0, 0};
    }
    __syncthreads();
#pragma unroll 8
    for (int nn = 0; nn < 32; ++nn) {
      cplx k = sk[tr][nn];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        cplx d = sd[tm + 16 * q][nn];
        acc[q].x += k.x * d.x - k.y * d.y;
        acc[q].y += k.x * d.y + k.y * d.x;
      }
    }
    __syncthreads();
  }
  if (r0 + tr < R)
    for (int q = 0; q < 4; ++q)
      if (m0 + tm + 16 * q < M) out[(size_t)(r0 + tr) * M + m0 + tm + 16 * q] = acc[q];
}

int g_dft_valu = 0;   // IA3_TUNE_DFT_VALU: 1 = the vector-unit contraction (tests compare the two)

// The same contraction on the matrix cores: out[r][m] = sum_n K[r][n] * in[m][n] is a complex (R x N) x (N x M) product
// — 15.7 Gflop for the first axis of a 50 x 512 x 512 crop — and the LDS-tiled kernel above spends five 16-byte LDS
// reads on every four complex multiply-adds.  v_mfma_f64_16x16x4_f64 takes one float64 of A and of B per lane
// (A[lane & 15][k = lane >> 4], B[k][lane & 15]) and leaves D[row = (lane >> 4) + 4 v][col = lane & 15] in four
// registers; a complex tile needs four real products (Kr Dr, Ki Di, Kr Di, Ki Dr).  A wave owns 16 rows r and 64
// columns m (four 16 x 16 tiles, 128 accumulator registers).  The sum over n does not care about order, so k slot
// q = lane >> 4 is given the run n0 + 8 q .. n0 + 8 q + 7 of a 32-sample chunk: every lane then reads whole 128-byte
// lines of K and of `in` straight from global memory (eight complex values each), no LDS, and eight MFMA steps consume
// them.  Float64 accumulation in the matrix unit: the upsampled peak search is as accurate as with the vector kernel.
typedef double v4d __attribute__((ext_vector_type(4)));
// Round 4, second form.  The first one loaded a tile's eight samples behind eight bounds branches, waited for them and
// only then issued its 32 matrix instructions: the loop alternated between a memory latency and 2 k cycles of MFMA, and
// two waves per SIMD reached 0.31 of the matrix peak.  Now the loads run ahead of the matrix unit in registers: every
// one of the four column tiles of a wave has a sample buffer of its own that is refilled for the NEXT chunk as soon as
// its 32 instructions of this chunk are issued (three tiles = 6 k cycles of matrix work before it is needed again), the
// K rows are double-buffered one chunk ahead, whole chunks are loaded without tests and the last, partial one through
// clamped addresses with the missing samples zeroed.  320 registers (128 of them accumulators): one wave per SIMD.
struct DftLd {
  const cplx* krow; const cplx* drow[4]; int N, kq;
  // eight samples of the whole chunk c for this lane's k slot: no tests, nothing computed on them until the matrix
  // instructions read them
  // Which of a chunk's 32 samples go to matrix step s and k slot q does not matter to the sum; n = 4 s + q makes the four
  // k slots of a row read 64 contiguous bytes per load instruction — 16 cache lines per instruction instead of 64 with
  // n = 8 q + s, which kept the vector L1 busier than the matrix unit (four waves x 40 loads x 64 lines against 8.2 k
  // cycles of MFMA per chunk)
  __device__ __forceinline__ void load(const cplx* row, int c, cplx (&v)[8]) const {
    const cplx* p = row + 32 * c + kq;
#pragma unroll
    for (int s = 0; s < 8; ++s) v[s] = p[4 * s];
  }
  // the last, partial chunk: clamped addresses, missing samples zero
  __device__ __forceinline__ void load_tail(const cplx* row, int c, cplx (&v)[8]) const {
    const int nb = 32 * c + kq;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int n = nb + 4 * s;
      const cplx x = row[n < N ? n : N - 1];
      v[s] = n < N ? x : cplx{0, 0};
    }
  }
};
__global__ __launch_bounds__(256, 1) void dft_contract_mfma_k(const cplx* __restrict__ K, const cplx* __restrict__ in,
                                                              cplx* __restrict__ out, int R, int M, int N, int ldk, int ldi) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int li = lane & 15, kq = lane >> 4;
  // The four waves of a block take two blocks of 16 rows x two groups of 64 columns: every `in` row a block reads is
  // read by two of its waves at the same time (one trip to HBM), so the R / 16 = 10 row blocks re-read `in` five times,
  // not ten — with four column groups per block the first contraction of a crop moved 1.05 GB and ran at the speed of
  // that (198 us; 107 us of matrix work)
  const int r0 = ((int)blockIdx.y * 2 + (wv & 1)) * 16, m0 = ((int)blockIdx.x * 2 + (wv >> 1)) * 64;
  if (m0 >= M || r0 >= R) return;                       // whole wave
  const int rr = r0 + li < R ? r0 + li : R - 1;         // rows past the end repeat the last one and are not stored
  DftLd L;
  L.krow = K + (size_t)rr * ldk; L.N = N; L.kq = kq;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int mm = m0 + 16 * t + li;
    L.drow[t] = in + (size_t)(mm < M ? mm : M - 1) * ldi;
  }
  v4d acc_rr[4], acc_ii[4], acc_ri[4], acc_ir[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) { acc_rr[t] = v4d{0, 0, 0, 0}; acc_ii[t] = acc_rr[t]; acc_ri[t] = acc_rr[t]; acc_ir[t] = acc_rr[t]; }
  const int nfull = N / 32;                             // whole chunks: pipelined
  cplx ka[8], kb[8], dv[4][8];
  auto tile = [&](const cplx (&kv)[8], int t) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      acc_rr[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(kv[s].x, dv[t][s].x, acc_rr[t], 0, 0, 0);
      acc_ii[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(kv[s].y, dv[t][s].y, acc_ii[t], 0, 0, 0);
      acc_ri[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(kv[s].x, dv[t][s].y, acc_ri[t], 0, 0, 0);
      acc_ir[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(kv[s].y, dv[t][s].x, acc_ir[t], 0, 0, 0);
    }
  };
  if (nfull > 0) {
    L.load(L.krow, 0, ka);
#pragma unroll
    for (int t = 0; t < 4; ++t) L.load(L.drow[t], 0, dv[t]);
    if (nfull > 1) L.load(L.krow, 1, kb);
    // four chunks per trip: the compiler drains every outstanding load where the loop closes (it cannot count loads
    // across the back edge), so the trips are long
#pragma unroll 1
    for (int c = 0; c < nfull; c += 4) {
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int cc = c + h;
        if (cc < nfull) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if (h % 2 == 0) tile(ka, t); else tile(kb, t);
            if (cc + 1 < nfull) L.load(L.drow[t], cc + 1, dv[t]);
          }
          if (cc + 2 < nfull) { if (h % 2 == 0) L.load(L.krow, cc + 2, ka); else L.load(L.krow, cc + 2, kb); }
        }
      }
    }
  }
  if (N % 32) {                                         // the partial chunk: all of its loads in flight together
    L.load_tail(L.krow, nfull, ka);
#pragma unroll
    for (int t = 0; t < 4; ++t) L.load_tail(L.drow[t], nfull, dv[t]);
#pragma unroll
    for (int t = 0; t < 4; ++t) tile(ka, t);
  }
  // D[row = kq + 4 v][col = li]: row -> r, col -> m
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int mm = m0 + 16 * t + li;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int r = r0 + kq + 4 * v;
      if (r < R && mm < M) out[(size_t)r * M + mm] = cplx{acc_rr[t][v] - acc_ii[t][v], acc_ri[t][v] + acc_ir[t][v]};
    }
  }
}

// (Round 4: a form that shares the `in` tile of five r blocks through a double-buffered LDS tile — 4.6x less L2 / MALL
// traffic, 246 registers, next tile prefetched into registers under the matrix instructions — was built and measured at
// 1.78-1.92 ms per alignment against 1.69 ms for the kernel above, bit-identical; it is not the re-reads that bound it.)

// ---- real-input form of the same computation -------------------------------------------------------------------
// Both stacks are real, so their spectra are Hermitian: D2Z transforms produce the half spectrum (Z, X, Yh = Y/2+1),
// half the work and half the traffic of Z2Z, and the cross-correlation comes back through Z2D as a real array.
template <class T>
__global__ void to_real_k(const T* __restrict__ a, double* __restrict__ o, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) o[i] = (double)a[i];
}
// One pass over the two half spectra: a := A conj(B) (optionally phase-normalised), b := conj(a) (the data of the
// upsampled DFT), and the block's share of sum |A|^2, sum |B|^2 over the FULL spectrum (columns 0 < ky < Y - ky count
// twice).  Fixed grid, partials added in index order afterwards: deterministic.
__global__ __launch_bounds__(256) void half_power_k(cplx* __restrict__ a, cplx* __restrict__ b, size_t n, int Yh, int Y,
                                                    int phase_norm, double* __restrict__ part_a, double* __restrict__ part_b) {
  __shared__ double sha[256], shb[256];
  double sa = 0, sb = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)ABS2_BLOCKS * 256) {
    const int ky = (int)(i % (size_t)Yh);
    const double wgt = (ky == 0 || 2 * ky == Y) ? 1.0 : 2.0;
    const cplx x = a[i], y = b[i];
    sa += wgt * (x.x * x.x + x.y * x.y);
    sb += wgt * (y.x * y.x + y.y * y.y);
    double re = x.x * y.x + x.y * y.y, im = x.y * y.x - x.x * y.y;
    if (phase_norm) {
      double m = hypot(re, im);
      const double lim = 100.0 * 2.220446049250313e-16;
      m = m > lim ? m : lim;
      re /= m; im /= m;
    }
    a[i] = cplx{re, im};
    b[i] = cplx{re, -im};
  }
  sha[threadIdx.x] = sa; shb[threadIdx.x] = sb;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) { sha[threadIdx.x] += sha[threadIdx.x + k]; shb[threadIdx.x] += shb[threadIdx.x + k]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part_a[blockIdx.x] = sha[0]; part_b[blockIdx.x] = shb[0]; }
}
__global__ __launch_bounds__(256) void real_argmax_part_k(const double* __restrict__ a, size_t n, double* pv, long long* pi) {
  __shared__ double sv[256];
  __shared__ long long si[256];
  double bv = -1.0; long long bi = 0x7fffffffffffffffLL;
  // four consecutive values per thread and step, two 16-byte loads in flight (one 8-byte load per step left the pass at
  // 3 TB/s); a thread still meets its values in ascending order, so the first of equal maxima wins as before
  typedef double d2 __attribute__((ext_vector_type(2)));
  const size_t n4 = n & ~(size_t)3;
  for (size_t k = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; k < n4; k += (size_t)gridDim.x * 1024) {
    const d2 p = *(const d2*)(a + k), q = *(const d2*)(a + k + 2);
    const double v[4] = {p.x * p.x, p.y * p.y, q.x * q.x, q.y * q.y};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (v[j] > bv) { bv = v[j]; bi = (long long)(k + j); }
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x < (unsigned)(n - n4)) {   // (the last n % 4 values)
    const size_t k = n4 + threadIdx.x;
    const double v = a[k] * a[k];
    if (v > bv || (v == bv && (long long)k < bi)) { bv = v; bi = (long long)k; }
  }
  sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      double ov = sv[threadIdx.x + s]; long long oi = si[threadIdx.x + s];
      if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { pv[blockIdx.x] = sv[0]; pi[blockIdx.x] = si[0]; }
}
// First contraction of the upsampled DFT from the half spectrum.  With D the (Hermitian) full data and K[r, Y - k] =
// conj(K[r, k]):   sum_{ky < Y} K[r, ky] D[kz, kx, ky] = P[r, kz, kx] + conj(P[r, -kz, -kx]) + K[r, 0] D[kz, kx, 0]
//                  (+ K[r, Y/2] D[kz, kx, Y/2] for even Y),   P[r, kz, kx] = sum_{0 < ky < Y - ky} K[r, ky] D[kz, kx, ky],
// so the matrix product runs over half the columns.  out / P: (R, Z, X); d: (Z, X, Yh); K: (R, ldk).
__global__ __launch_bounds__(256) void half_combine_k(const cplx* __restrict__ P, const cplx* __restrict__ d,
                                                      const cplx* __restrict__ K, int ldk, int R, int Z, int X, int Yh, int Y,
                                                      cplx* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t zx = (size_t)Z * X;
  if (i >= (size_t)R * zx) return;
  const int r = (int)(i / zx);
  const size_t m = i % zx;
  const int kz = (int)(m / X), kx = (int)(m % X);
  const size_t mm = (size_t)((Z - kz) % Z) * X + (size_t)((X - kx) % X);
  const cplx p = P[i], q = P[(size_t)r * zx + mm];
  const cplx k0 = K[(size_t)r * ldk], d0 = d[m * Yh];
  double re = p.x + q.x + (k0.x * d0.x - k0.y * d0.y);
  double im = p.y - q.y + (k0.x * d0.y + k0.y * d0.x);
  if (2 * (Yh - 1) == Y) {   // even Y: the Nyquist column stands alone as well
    const cplx kn = K[(size_t)r * ldk + Yh - 1], dn = d[m * Yh + Yh - 1];
    re += kn.x * dn.x - kn.y * dn.y;
    im += kn.x * dn.y + kn.y * dn.x;
  }
  out[i] = cplx{re, im};
}

// Peak of a correlation + the few scalars the host needs with it, in ONE wait: the partial maxima are reduced by a
// one-block kernel that also picks the value at the peak and the two power sums and writes everything into the thread's
// pinned mailbox, followed by the sequence word the host polls.  (Four device-to-host copies with their synchronisations
// stood between the coarse peak and the upsampled DFT and three more behind it: ~285 us of idle device per crop in the
// kernel trace, a fifth of align_image.)  Same choice as the host loop of abs_argmax / real_argmax: the largest value,
// the smallest index among equals; no finite value at all gives index 0.
struct PeakMail { unsigned seq, pad; long long idx; double val2, re, im, sum0, sum1; };
__global__ __launch_bounds__(256) void argmax_final_k(const double* __restrict__ pv, const long long* __restrict__ pi, int nb,
                                                      const double* __restrict__ real_src, const cplx* __restrict__ cplx_src,
                                                      const double* __restrict__ sums, volatile PeakMail* mail, unsigned seq) {
  __shared__ double sv[256];
  __shared__ long long si[256];
  double bv = -1.0; long long bi = 0x7fffffffffffffffLL;
  for (int k = threadIdx.x; k < nb; k += 256) {
    const double v = pv[k]; const long long i = pi[k];
    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
  }
  sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const double ov = sv[threadIdx.x + s]; const long long oi = si[threadIdx.x + s];
      if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    long long idx = si[0];
    double v2 = sv[0];
    if (!(v2 >= 0.0)) { idx = 0; v2 = -1.0; }   // nothing compared greater than the start value (all NaN): the host loop's answer
    mail->idx = idx; mail->val2 = v2;
    mail->re = real_src ? real_src[idx] : (cplx_src ? cplx_src[idx].x : 0.0);
    mail->im = cplx_src ? cplx_src[idx].y : 0.0;
    mail->sum0 = sums ? sums[0] : 0.0; mail->sum1 = sums ? sums[1] : 0.0;
    __threadfence_system();
    mail->seq = seq;
  }
}
// partials of a |.|^2 argmax already queued on the stream -> peak record on the host
static int peak_to_host(const double* pv, const long long* pi, int nb, const double* real_src, const cplx* cplx_src,
                        const double* sums, PeakMail* out) {
  hipStream_t st = stream();
  void *mh = nullptr, *md = nullptr;
  constexpr size_t OFF = 1024;   // the seed stage uses the first 64 bytes of the mailbox, the fit 2048 and up
  if (host_mailbox(4096, &mh, &md) != IA3_OK) return IA3_ENOMEM;
  static thread_local unsigned t_seq = 0;
  const unsigned seq = ++t_seq ? t_seq : ++t_seq;
  hipLaunchKernelGGL(argmax_final_k, dim3(1), dim3(256), 0, st, pv, pi, nb, real_src, cplx_src, sums,
                     (volatile PeakMail*)((char*)md + OFF), seq);
  IA3_KCHECK();
  volatile PeakMail* mb = (volatile PeakMail*)((char*)mh + OFF);
  SpinWait sw;
  while (mb->seq != seq) {
    sw.relax();
    if (((sw.n & 0xfffff) == 0 || (sw.n > 40400 && (sw.n & 0x3ff) == 0)) && hipStreamQuery(st) != hipErrorNotReady) {   // drained (or failed) without the word
      if (mb->seq == seq) break;
      IA3_HIP(hipStreamSynchronize(st));
      if (mb->seq != seq) return set_error(IA3_EHIP, "correlation peak did not reach the host mailbox");
      break;
    }
  }
  out->idx = mb->idx; out->val2 = mb->val2; out->re = mb->re; out->im = mb->im; out->sum0 = mb->sum0; out->sum1 = mb->sum1;
  return IA3_OK;
}
int real_peak(const double* a, size_t n, const double* sums, PeakMail* out) {
  hipStream_t st = stream();
  const int nb = 512;
  Scratch pv(nb * sizeof(double)), pi(nb * sizeof(long long));
  if (!pv.p || !pi.p) return IA3_ENOMEM;
  hipLaunchKernelGGL(real_argmax_part_k, dim3(nb), dim3(256), 0, st, a, n, pv.as<double>(), pi.as<long long>());
  return peak_to_host(pv.as<double>(), pi.as<long long>(), nb, a, nullptr, sums, out);
}
int abs_peak(const cplx* a, size_t n, const double* sums, PeakMail* out) {
  hipStream_t st = stream();
  const int nb = 512;
  Scratch pv(nb * sizeof(double)), pi(nb * sizeof(long long));
  if (!pv.p || !pi.p) return IA3_ENOMEM;
  hipLaunchKernelGGL(abs_argmax_part_k, dim3(nb), dim3(256), 0, st, a, n, pv.as<double>(), pi.as<long long>());
  return peak_to_host(pv.as<double>(), pi.as<long long>(), nb, nullptr, a, sums, out);
}

int g_fft_c2c = 0;   // IA3_TUNE_FFT_C2C: 1 = complex transforms of the real stacks (first version; tests compare)

}  // namespace

namespace ia3k { void set_dft_valu(int on) { g_dft_valu = on != 0; } void set_fft_c2c(int on) { g_fft_c2c = on != 0; } }

extern "C" {

int ia3_fftalign_2d(const double* im1, int s1x, int s1y, const double* im2, int s2x, int s2y,
                    const double* center, double max_disp, int* out_xy) {
  int rc = ensure_init(); if (rc) return rc;
  if (!im1 || !im2 || !center || !out_xy) return set_error(IA3_EINVAL, "null argument");
  hipStream_t st = stream();
  Scratch a((size_t)s1x * s1y * sizeof(double)), b((size_t)s2x * s2y * sizeof(double));
  if (!a.p || !b.p) return IA3_ENOMEM;
  IA3_HIP(hipMemcpyAsync(a.p, im1, (size_t)s1x * s1y * sizeof(double), hipMemcpyHostToDevice, st));
  IA3_HIP(hipMemcpyAsync(b.p, im2, (size_t)s2x * s2y * sizeof(double), hipMemcpyHostToDevice, st));
  return fftalign2d_dev(a.as<double>(), s1x, s1y, b.as<double>(), s2x, s2y, center, max_disp, out_xy);
}

// alignment_tools.py:330-353 with gb <= 1 (the production default fft_filt_size=0, alignment.py:141,191-193)
int ia3_fft3d_from2d_dev(const ia3_stack* im1, const ia3_stack* im2, double max_disp, int* out_zxy) {
  int rc = ensure_init(); if (rc) return rc;
  if (!im1 || !im2 || !out_zxy) return set_error(IA3_EINVAL, "null argument");
  if (im1->dtype != im2->dtype) return set_error(IA3_EINVAL, "stacks differ in dtype");
  hipStream_t st = stream();
  const int sx = im1->X, sy = im1->Y;
  const double center[2] = {0, 0};
  int txy[2], tzq[2];
  {
    Scratch p1((size_t)im1->X * im1->Y * sizeof(double)), p2((size_t)im2->X * im2->Y * sizeof(double));
    if (!p1.p || !p2.p) return IA3_ENOMEM;
    {
      ProfScope ps("maxproj_z");
      if (im1->dtype == IA3_F32) { launch_maxproj_z<float>(im1, 0, im1->X, 0, im1->Y, p1.as<double>(), st); launch_maxproj_z<float>(im2, 0, im2->X, 0, im2->Y, p2.as<double>(), st); }
      else { launch_maxproj_z<uint16_t>(im1, 0, im1->X, 0, im1->Y, p1.as<double>(), st); launch_maxproj_z<uint16_t>(im2, 0, im2->X, 0, im2->Y, p2.as<double>(), st); }
    }
    rc = fftalign2d_dev(p1.as<double>(), im1->X, im1->Y, p2.as<double>(), im2->X, im2->Y, center, max_disp, txy);
    if (rc) return rc;
  }
  const int tx = txy[0], ty = txy[1];
  // im1[:, max(tx,0):sx+tx, max(ty,0):sy+ty] and im2[:, max(-tx,0):sx-tx, max(-ty,0):sy-ty] (python slice clipping)
  auto clip = [](int v, int n) { return v < 0 ? 0 : (v > n ? n : v); };
  const int a0 = clip(tx > 0 ? tx : 0, im1->X), a1 = clip(sx + tx, im1->X), b0 = clip(ty > 0 ? ty : 0, im1->Y), b1 = clip(sy + ty, im1->Y);
  const int c0 = clip(-tx > 0 ? -tx : 0, im2->X), c1 = clip(sx - tx, im2->X), d0 = clip(-ty > 0 ? -ty : 0, im2->Y), d1 = clip(sy - ty, im2->Y);
  const int n1x = a1 - a0, n1y = b1 - b0, n2x = c1 - c0, n2y = d1 - d0;
  if (n1x < 1 || n1y < 1 || n2x < 1 || n2y < 1) return set_error(IA3_EINVAL, "xy shift (%d,%d) leaves no overlap", tx, ty);
  {
    Scratch q1((size_t)im1->Z * n1x * sizeof(double)), q2((size_t)im2->Z * n2x * sizeof(double));
    if (!q1.p || !q2.p) return IA3_ENOMEM;
    {
      ProfScope ps("maxproj_y");
      if (im1->dtype == IA3_F32) { launch_maxproj_y<float>(im1, a0, n1x, b0, n1y, q1.as<double>(), st); launch_maxproj_y<float>(im2, c0, n2x, d0, n2y, q2.as<double>(), st); }
      else { launch_maxproj_y<uint16_t>(im1, a0, n1x, b0, n1y, q1.as<double>(), st); launch_maxproj_y<uint16_t>(im2, c0, n2x, d0, n2y, q2.as<double>(), st); }
    }
    rc = fftalign2d_dev(q1.as<double>(), im1->Z, n1x, q2.as<double>(), im2->Z, n2x, center, max_disp, tzq);
    if (rc) return rc;
  }
  out_zxy[0] = tzq[0]; out_zxy[1] = tx; out_zxy[2] = ty;
  return IA3_OK;
}

int ia3_fft3d_from2d(const void* im1, const void* im2, int dtype, int Z, int X, int Y, double max_disp, int* out_zxy) {
  ia3_stack *a = nullptr, *b = nullptr;
  int rc = ia3_stack_upload(im1, dtype, Z, X, Y, &a); if (rc) return rc;
  rc = ia3_stack_upload(im2, dtype, Z, X, Y, &b);
  if (!rc) rc = ia3_fft3d_from2d_dev(a, b, max_disp, out_zxy);
  ia3_stack_free(a); ia3_stack_free(b);
  return rc;
}

// phase_cross_correlation(reference, moving, upsample_factor, normalization) -> shift, error, phasediff
int ia3_phase_xcorr3d_dev(const ia3_stack* ref, const ia3_stack* mov, int upsample, int normalization,
                          double* shift, double* err, double* phasediff) {
  int rc = ensure_init(); if (rc) return rc;
  if (!ref || !mov || !shift) return set_error(IA3_EINVAL, "null argument");
  if (ref->dtype != mov->dtype || ref->Z != mov->Z || ref->X != mov->X || ref->Y != mov->Y)
    return set_error(IA3_EINVAL, "reference and moving stacks differ in shape or dtype");
  if (upsample < 1) return set_error(IA3_EINVAL, "upsample_factor must be >= 1");
  hipStream_t st = stream();
  const int Z = ref->Z, X = ref->X, Y = ref->Y;
  const size_t n = (size_t)Z * X * Y;
  const int Yh = Y / 2 + 1;
  const size_t nh = (size_t)Z * X * Yh;
  const bool half = !g_fft_c2c && Y >= 8;   // real-input transforms (half spectra); tiny rows keep the complex form
  Scratch fa((half ? nh : n) * sizeof(cplx)), fb((half ? nh : n) * sizeof(cplx)), sums(2 * sizeof(double)),
      parts(2 * ABS2_BLOCKS * sizeof(double)), rbuf(half ? n * sizeof(double) : 256);
  if (!fa.p || !fb.p || !sums.p || !parts.p || !rbuf.p) return IA3_ENOMEM;
  const unsigned nb = (unsigned)((n + 255) / 256);
  ProfScope ps("phase_xcorr3d");
  long long idx; double v2;
  cplx ccmax;
  double hs[2];
  if (half) {
    PlanLease fwd, inv;   // released at the end of this block: real_argmax below synchronises the stream first
    { int prc = get_plan(HIPFFT_D2Z, Z, X, Y, st, fwd); if (prc) return prc; }
    { int prc = get_plan(HIPFFT_Z2D, Z, X, Y, st, inv); if (prc) return prc; }
    for (int which = 0; which < 2; ++which) {
      const ia3_stack* s = which ? mov : ref;
      if (s->dtype == IA3_F32) hipLaunchKernelGGL((to_real_k<float>), dim3(nb), dim3(256), 0, st, (const float*)s->d, rbuf.as<double>(), n);
      else hipLaunchKernelGGL((to_real_k<uint16_t>), dim3(nb), dim3(256), 0, st, (const uint16_t*)s->d, rbuf.as<double>(), n);
      IA3_FFT(hipfftExecD2Z(fwd.h, rbuf.as<double>(), which ? fb.as<cplx>() : fa.as<cplx>()));
    }
    // fa := prod, fb := conj(prod), power sums of both spectra — one pass
    hipLaunchKernelGGL(half_power_k, dim3(ABS2_BLOCKS), dim3(256), 0, st, fa.as<cplx>(), fb.as<cplx>(), nh, Yh, Y, normalization,
                       parts.as<double>(), parts.as<double>() + ABS2_BLOCKS);
    hipLaunchKernelGGL(abs2_sum_k, dim3(1), dim3(1024), 0, st, (const double*)parts.as<double>(), sums.as<double>());
    hipLaunchKernelGGL(abs2_sum_k, dim3(1), dim3(1024), 0, st, (const double*)(parts.as<double>() + ABS2_BLOCKS), sums.as<double>() + 1);
    // rbuf := ifftn(prod), real (unnormalised by hipFFT; the transform may overwrite fa)
    IA3_FFT(hipfftExecZ2D(inv.h, fa.as<cplx>(), rbuf.as<double>()));
    IA3_KCHECK();
    PeakMail pk;
    rc = real_peak(rbuf.as<double>(), n, sums.as<double>(), &pk); if (rc) return rc;
    idx = pk.idx; v2 = pk.val2; hs[0] = pk.sum0; hs[1] = pk.sum1;
    ccmax = cplx{pk.re / (double)n, 0.0};
  } else {
    if (ref->dtype == IA3_F32) {
      hipLaunchKernelGGL((to_cplx_k<float>), dim3(nb), dim3(256), 0, st, (const float*)ref->d, fa.as<cplx>(), n);
      hipLaunchKernelGGL((to_cplx_k<float>), dim3(nb), dim3(256), 0, st, (const float*)mov->d, fb.as<cplx>(), n);
    } else {
      hipLaunchKernelGGL((to_cplx_k<uint16_t>), dim3(nb), dim3(256), 0, st, (const uint16_t*)ref->d, fa.as<cplx>(), n);
      hipLaunchKernelGGL((to_cplx_k<uint16_t>), dim3(nb), dim3(256), 0, st, (const uint16_t*)mov->d, fb.as<cplx>(), n);
    }
    PlanLease plan;
    { int prc = get_plan(HIPFFT_Z2Z, Z, X, Y, st, plan); if (prc) return prc; }
    IA3_FFT(hipfftExecZ2Z(plan.h, fa.as<cplx>(), fa.as<cplx>(), HIPFFT_FORWARD));
    IA3_FFT(hipfftExecZ2Z(plan.h, fb.as<cplx>(), fb.as<cplx>(), HIPFFT_FORWARD));
    hipLaunchKernelGGL(abs2_part_k, dim3(ABS2_BLOCKS), dim3(256), 0, st, (const cplx*)fa.as<cplx>(), n, parts.as<double>());
    hipLaunchKernelGGL(abs2_sum_k, dim3(1), dim3(1024), 0, st, (const double*)parts.as<double>(), sums.as<double>());
    hipLaunchKernelGGL(abs2_part_k, dim3(ABS2_BLOCKS), dim3(256), 0, st, (const cplx*)fb.as<cplx>(), n, parts.as<double>());
    hipLaunchKernelGGL(abs2_sum_k, dim3(1), dim3(1024), 0, st, (const double*)parts.as<double>(), sums.as<double>() + 1);
    hipLaunchKernelGGL(cross_power_k, dim3(nb), dim3(256), 0, st, fa.as<cplx>(), (const cplx*)fb.as<cplx>(), n, normalization);
    // fb := ifftn(prod) (unnormalised by hipFFT: scale 1/n applied to the picked value only)
    IA3_FFT(hipfftExecZ2Z(plan.h, fa.as<cplx>(), fb.as<cplx>(), HIPFFT_BACKWARD));
    IA3_KCHECK();
    PeakMail pk;
    rc = abs_peak(fb.as<cplx>(), n, sums.as<double>(), &pk); if (rc) return rc;
    idx = pk.idx; v2 = pk.val2; hs[0] = pk.sum0; hs[1] = pk.sum1;
    ccmax = cplx{pk.re / (double)n, pk.im / (double)n};
  }
  const int dims[3] = {Z, X, Y};
  long long rem = idx;
  int peak[3];
  peak[2] = (int)(rem % Y); rem /= Y; peak[1] = (int)(rem % X); rem /= X; peak[0] = (int)rem;
  double sh[3];
  for (int a = 0; a < 3; ++a) {
    double mid = (double)(dims[a] / 2);  // np.fix(size/2)
    sh[a] = peak[a];
    if (sh[a] > mid) sh[a] -= dims[a];
  }
  double src_amp = hs[0], tgt_amp = hs[1];
  if (upsample == 1) {
    src_amp /= (double)n; tgt_amp /= (double)n;
  } else {
    const double u = (double)upsample;
    for (int a = 0; a < 3; ++a) sh[a] = nearbyint(sh[a] * u) / u;
    const int R = (int)ceil(u * 1.5);
    const double dftshift = trunc(R / 2.0);
    double off[3];
    for (int a = 0; a < 3; ++a) off[a] = dftshift - sh[a] * u;
    // data = conj(prod); contract last axis three times (axes Y, X, Z), new axis goes first
    if (!half) hipLaunchKernelGGL(conj_k, dim3(nb), dim3(256), 0, st, fa.as<cplx>(), n);
    const int maxN = Y > X ? (Y > Z ? Y : Z) : (X > Z ? X : Z);
    Scratch K((size_t)R * maxN * sizeof(cplx)), t1((size_t)R * Z * X * sizeof(cplx)), t2((size_t)R * R * Z * sizeof(cplx)),
        t3((size_t)R * R * R * sizeof(cplx)), tp(half ? (size_t)R * Z * X * sizeof(cplx) : 256);
    if (!K.p || !t1.p || !t2.p || !t3.p || !tp.p) return IA3_ENOMEM;
    // out[r][m] = sum_{n < cols} Kmat[r][n] in[m][n], rows of Kmat / in ldk / ldi elements apart
    auto matmul = [&](const cplx* Kmat, const cplx* in, cplx* out, int M, int cols, int ldk, int ldi) {
      if (g_dft_valu)
        hipLaunchKernelGGL(dft_contract_k, dim3((M + 63) / 64, (R + 15) / 16), dim3(256), 0, st, Kmat, in, out, R, M, cols, ldk, ldi);
      else
        hipLaunchKernelGGL(dft_contract_mfma_k, dim3((M + 127) / 128, (R + 31) / 32), dim3(256), 0, st, Kmat, in, out, R, M, cols, ldk, ldi);
    };
    auto contract = [&](const cplx* in, cplx* out, int M, int N, double o) {
      hipLaunchKernelGGL(dft_kernel_k, dim3((N + 255) / 256, R), dim3(256), 0, st, K.as<cplx>(), R, N, o, u);
      matmul((const cplx*)K.as<cplx>(), in, out, M, N, N, N);
    };
    if (half) {
      // (R_y, Z, X) from the half spectrum in fb: product over the interior columns, then the Hermitian completion
      hipLaunchKernelGGL(dft_kernel_k, dim3((Y + 255) / 256, R), dim3(256), 0, st, K.as<cplx>(), R, Y, off[2], u);
      const int inner = (Y - 1) / 2;   // columns 1 .. inner have a distinct mirror image
      matmul((const cplx*)K.as<cplx>() + 1, (const cplx*)fb.as<cplx>() + 1, tp.as<cplx>(), Z * X, inner, Y, Yh);
      hipLaunchKernelGGL(half_combine_k, dim3((unsigned)(((size_t)R * Z * X + 255) / 256)), dim3(256), 0, st, (const cplx*)tp.as<cplx>(),
                         (const cplx*)fb.as<cplx>(), (const cplx*)K.as<cplx>(), Y, R, Z, X, Yh, Y, t1.as<cplx>());
    } else {
      contract(fa.as<cplx>(), t1.as<cplx>(), Z * X, Y, off[2]);   // (R_y, Z, X)
    }
    contract(t1.as<cplx>(), t2.as<cplx>(), R * Z, X, off[1]);   // (R_x, R_y, Z)
    contract(t2.as<cplx>(), t3.as<cplx>(), R * R, Z, off[0]);   // (R_z, R_x, R_y)
    IA3_KCHECK();
    const size_t nr = (size_t)R * R * R;
    PeakMail fine;
    rc = abs_peak(t3.as<cplx>(), nr, nullptr, &fine); if (rc) return rc;   // |conj(x)| = |x|
    idx = fine.idx; v2 = fine.val2;
    ccmax = cplx{fine.re, -fine.im};  // .conj()
    rem = idx;
    int pk[3];
    pk[2] = (int)(rem % R); rem /= R; pk[1] = (int)(rem % R); rem /= R; pk[0] = (int)rem;
    for (int a = 0; a < 3; ++a) sh[a] += ((double)pk[a] - dftshift) / u;
  }
  for (int a = 0; a < 3; ++a) { if (dims[a] == 1) sh[a] = 0; shift[a] = sh[a]; }
  const double amp = src_amp * tgt_amp;
  if (err) *err = amp != 0 ? sqrt(fabs(1.0 - (ccmax.x * ccmax.x + ccmax.y * ccmax.y) / amp)) : NAN;
  if (phasediff) *phasediff = atan2(ccmax.y, ccmax.x);
  return IA3_OK;
}

}  // extern "C"  (reopened below)

// ---- align_image's crop loop without host round trips inside a crop ------------------------------------------------
// correction_tools/alignment.py:617-662 runs phase_cross_correlation(ref crop, src crop) crop after crop, and every
// image of a run is aligned to the SAME reference bead image (classes/batch_functions.py:169-206).  Three things follow:
//  * the reference crop's half spectrum is a run constant: a DriftRef keeps it (105 MB per 50 x 512 x 512 crop) and a crop
//    costs two transforms instead of three;
//  * a crop is read out of the full stack straight into the transform's float64 input (crop_to_real_k): no crop copies;
//  * the coarse peak never goes to the host: a one-block kernel turns it into the offsets of the upsampled DFT's kernel
//    matrices, so the whole chain of a crop — D2Z, cross-power, Z2D, argmax, three contractions, argmax — is queued at
//    once, and the first crops of an image (align_image needs three anyway) are queued back to back with ONE wait for
//    their three shifts.  (ia3_phase_xcorr3d_dev waits twice per crop: ~0.1 ms of idle device each time.)
// Same arithmetic as ia3_phase_xcorr3d_dev on the same operands; shifts agree to the last bit (tests).
namespace {

// (a row of the crop per block row: no 64-bit divisions per value, two values per thread)
template <class T>
__global__ __launch_bounds__(256) void crop_to_real_k(const T* __restrict__ im, int X, int Y, int z0, int x0, int y0, int cz, int cx, int cy,
                                                      double* __restrict__ o) {
  const int x = (int)blockIdx.y, z = (int)blockIdx.z;
  const size_t r = (size_t)z * cx + x;                    // row of the crop
  const T* src = im + ((size_t)(z0 + z) * X + (size_t)(x0 + x)) * Y + (size_t)y0;
  double* dst = o + (size_t)r * cy;
  const int y = (int)(blockIdx.x * 256 + threadIdx.x) * 2;
  if (y + 1 < cy) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2 v = {(double)src[y], (double)src[y + 1]};
    if ((((size_t)r * cy + y) & 1) == 0) *(d2*)(dst + y) = v;   // 16-byte aligned
    else { dst[y] = v.x; dst[y + 1] = v.y; }
  } else if (y < cy) dst[y] = (double)src[y];
}
// out-of-place form of half_power_k: prod := A conj(B) (optionally phase-normalised), cj := conj(prod); A stays
__global__ __launch_bounds__(256) void half_power2_k(const cplx* __restrict__ a, const cplx* __restrict__ b, size_t n, int Yh, int Y,
                                                     int phase_norm, cplx* __restrict__ prod, cplx* __restrict__ cj,
                                                     double* __restrict__ part_a, double* __restrict__ part_b) {
  __shared__ double sha[256], shb[256];
  double sa = 0, sb = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)ABS2_BLOCKS * 256) {
    const int ky = (int)(i % (size_t)Yh);
    const double wgt = (ky == 0 || 2 * ky == Y) ? 1.0 : 2.0;
    const cplx x = a[i], y = b[i];
    sa += wgt * (x.x * x.x + x.y * x.y);
    sb += wgt * (y.x * y.x + y.y * y.y);
    double re = x.x * y.x + x.y * y.y, im = x.y * y.x - x.x * y.y;
    if (phase_norm) {
      double m = hypot(re, im);
      const double lim = 100.0 * 2.220446049250313e-16;
      m = m > lim ? m : lim;
      re /= m; im /= m;
    }
    prod[i] = cplx{re, im};
    cj[i] = cplx{re, -im};
  }
  sha[threadIdx.x] = sa; shb[threadIdx.x] = sb;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) { sha[threadIdx.x] += sha[threadIdx.x + k]; shb[threadIdx.x] += shb[threadIdx.x + k]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part_a[blockIdx.x] = sha[0]; part_b[blockIdx.x] = shb[0]; }
}
// what the host does between the coarse peak and the upsampled DFT (ia3_phase_xcorr3d_dev), on the device
struct CoarseRec { double sh[3]; double off[3]; };
__global__ __launch_bounds__(256) void coarse_final_k(const double* __restrict__ pv, const long long* __restrict__ pi, int nb,
                                                      int Z, int X, int Y, double u, double dftshift, CoarseRec* rec) {
  __shared__ double sv[256];
  __shared__ long long si[256];
  double bv = -1.0; long long bi = 0x7fffffffffffffffLL;
  for (int k = threadIdx.x; k < nb; k += 256) {
    const double v = pv[k]; const long long i = pi[k];
    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
  }
  sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const double ov = sv[threadIdx.x + s]; const long long oi = si[threadIdx.x + s];
      if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    long long idx = si[0];
    if (!(sv[0] >= 0.0)) idx = 0;
    const int dims[3] = {Z, X, Y};
    long long rem = idx;
    int peak[3];
    peak[2] = (int)(rem % Y); rem /= Y; peak[1] = (int)(rem % X); rem /= X; peak[0] = (int)rem;
    for (int a = 0; a < 3; ++a) {
      double sh = (double)peak[a];
      if (sh > (double)(dims[a] / 2)) sh -= (double)dims[a];
      sh = nearbyint(sh * u) / u;
      rec->sh[a] = sh;
      rec->off[a] = dftshift - sh * u;
    }
  }
}
// dft_kernel_k with the offset read from the coarse record
__global__ void dft_kernel_dev_k(cplx* __restrict__ K, int R, int N, const double* __restrict__ off, double u) {
  int n = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (n >= N) return;
  int kf = n < (N + 1) / 2 ? n : n - N;
  double turns = ((double)r - *off) * ((double)kf / ((double)N * u));
  double s, c;
  sincospi(-2.0 * turns, &s, &c);
  K[(size_t)r * N + n] = cplx{c, s};
}
struct ShiftMail { unsigned seq, pad; double shift[3]; };
__global__ __launch_bounds__(256) void fine_final_k(const double* __restrict__ pv, const long long* __restrict__ pi, int nb, int R,
                                                    double u, double dftshift, const CoarseRec* __restrict__ rec, int Z, int X, int Y,
                                                    volatile ShiftMail* mail, unsigned seq) {
  __shared__ double sv[256];
  __shared__ long long si[256];
  double bv = -1.0; long long bi = 0x7fffffffffffffffLL;
  for (int k = threadIdx.x; k < nb; k += 256) {
    const double v = pv[k]; const long long i = pi[k];
    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
  }
  sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const double ov = sv[threadIdx.x + s]; const long long oi = si[threadIdx.x + s];
      if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    long long idx = si[0];
    if (!(sv[0] >= 0.0)) idx = 0;
    long long rem = idx;
    int pk[3];
    pk[2] = (int)(rem % R); rem /= R; pk[1] = (int)(rem % R); rem /= R; pk[0] = (int)rem;
    const int dims[3] = {Z, X, Y};
    for (int a = 0; a < 3; ++a) {
      double sh = rec->sh[a] + ((double)pk[a] - dftshift) / u;
      if (dims[a] == 1) sh = 0;
      mail->shift[a] = sh;
    }
    __threadfence_system();
    mail->seq = seq;
  }
}

}  // namespace

namespace ia3k {

struct DriftRef {
  const ia3_stack* ref = nullptr;
  int dtype = 0, n = 0;
  int box[8][6] = {};
  void* spec[8] = {};       // half spectrum of the reference crop
  bool have[8] = {};
  bool cached[8] = {};      // spec[i] is a block of the scratch cache (a temporary reference) rather than a hipMalloc
  std::mutex mu;            // a lazily made spectrum (temporary refs are used by one thread; shared ones are made eagerly)
};

static int crop_spectrum(const ia3_stack* s, const int* b, cplx* out, double* rbuf, hipStream_t st) {
  const int cz = b[1] - b[0], cx = b[3] - b[2], cy = b[5] - b[4];
  const size_t n = (size_t)cz * cx * cy;
  const unsigned nb = (unsigned)((n + 255) / 256);
  if (s->dtype == IA3_F32) hipLaunchKernelGGL((crop_to_real_k<float>), dim3((unsigned)((cy + 511) / 512), (unsigned)cx, (unsigned)cz), dim3(256), 0, st, (const float*)s->d, s->X, s->Y, b[0], b[2], b[4], cz, cx, cy, rbuf);
  else hipLaunchKernelGGL((crop_to_real_k<uint16_t>), dim3((unsigned)((cy + 511) / 512), (unsigned)cx, (unsigned)cz), dim3(256), 0, st, (const uint16_t*)s->d, s->X, s->Y, b[0], b[2], b[4], cz, cx, cy, rbuf);
  PlanLease fwd;
  { int prc = get_plan(HIPFFT_D2Z, cz, cx, cy, st, fwd); if (prc) return prc; }
  IA3_FFT(hipfftExecD2Z(fwd.h, rbuf, out));
  IA3_KCHECK();
  IA3_HIP(hipStreamSynchronize(st));   // the plan goes back to the pool when this returns
  return IA3_OK;
}

void drift_ref_free(DriftRef* r) {
  if (!r) return;
  for (int i = 0; i < 8; ++i)
    if (r->spec[i]) { if (r->cached[i]) ws_put(r->spec[i]); else (void)hipFree(r->spec[i]); }
  delete r;
}

int drift_ref_create(const ia3_stack* ref, const int* crops, int n_crops, bool eager, DriftRef** out) {
  int rc = ensure_init(); if (rc) return rc;
  if (!ref || !crops || !out || n_crops < 1 || n_crops > 8) return set_error(IA3_EINVAL, "bad drift reference arguments");
  DriftRef* r = new DriftRef();
  r->ref = ref; r->dtype = ref->dtype; r->n = n_crops;
  for (int i = 0; i < n_crops; ++i) {
    const int* c = crops + 6 * i;
    if (c[0] < 0 || c[2] < 0 || c[4] < 0 || c[1] > ref->Z || c[3] > ref->X || c[5] > ref->Y || c[0] >= c[1] || c[2] >= c[3] || c[4] >= c[5]) {
      drift_ref_free(r);
      return set_error(IA3_EINVAL, "bad crop [%d:%d, %d:%d, %d:%d] of (%d,%d,%d)", c[0], c[1], c[2], c[3], c[4], c[5], ref->Z, ref->X, ref->Y);
    }
    for (int k = 0; k < 6; ++k) r->box[i][k] = c[k];
  }
  if (eager) {
    hipStream_t st = stream();
    for (int i = 0; i < n_crops && !rc; ++i) {
      const int* b = r->box[i];
      const int cz = b[1] - b[0], cx = b[3] - b[2], cy = b[5] - b[4];
      if (cy < 8) continue;   // such crops take the complex-transform route (no cached spectrum)
      const size_t n = (size_t)cz * cx * cy, nh = (size_t)cz * cx * (cy / 2 + 1);
      if (hipMalloc(&r->spec[i], nh * sizeof(cplx)) != hipSuccess) { rc = set_error(IA3_ENOMEM, "reference spectrum of crop %d", i); break; }
      Scratch rbuf(n * sizeof(double));
      if (!rbuf.p) { rc = IA3_ENOMEM; break; }
      rc = crop_spectrum(ref, b, (cplx*)r->spec[i], rbuf.as<double>(), st);
      r->have[i] = !rc;
    }
    if (rc) { drift_ref_free(r); return rc; }
  }
  *out = r;
  return IA3_OK;
}

// shifts of crops [first, first + count) of `src` against the reference; count <= 3; one wait for all of them
int drift_crops(const ia3_stack* src, DriftRef* ref, int first, int count, int upsample, int normalization, double* shifts) {
  int rc = ensure_init(); if (rc) return rc;
  if (!src || !ref || !shifts || first < 0 || count < 1 || count > 3 || first + count > ref->n) return set_error(IA3_EINVAL, "bad crop range");
  if (src->dtype != ref->dtype || src->Z != ref->ref->Z || src->X != ref->ref->X || src->Y != ref->ref->Y)
    return set_error(IA3_EINVAL, "source and reference stacks differ in shape or dtype");
  hipStream_t st = stream();
  bool fast = !g_fft_c2c && upsample > 1;
  for (int c = first; c < first + count; ++c) {
    const int* b = ref->box[c];
    if (b[5] - b[4] < 8) fast = false;
    if (c > first && (b[1] - b[0] != ref->box[first][1] - ref->box[first][0] || b[3] - b[2] != ref->box[first][3] - ref->box[first][2] ||
                      b[5] - b[4] != ref->box[first][5] - ref->box[first][4])) fast = false;   // (one set of buffers and plans per batch)
  }
  if (!fast) {   // the general route, crop by crop (complex transforms, upsample 1, tiny rows, crops of different sizes)
    for (int c = first; c < first + count; ++c) {
      const int* b = ref->box[c];
      ia3_stack *sc = nullptr, *rcp = nullptr;
      rc = ia3_stack_crop(src, b[0], b[1], b[2], b[3], b[4], b[5], &sc);
      if (!rc) rc = ia3_stack_crop(ref->ref, b[0], b[1], b[2], b[3], b[4], b[5], &rcp);
      if (!rc) rc = ia3_phase_xcorr3d_dev(rcp, sc, upsample, normalization, shifts + 3 * (c - first), nullptr, nullptr);
      if (sc) ia3_stack_free(sc);
      if (rcp) ia3_stack_free(rcp);
      if (rc) return rc;
    }
    return IA3_OK;
  }
  const int* b0 = ref->box[first];
  const int Z = b0[1] - b0[0], X = b0[3] - b0[2], Y = b0[5] - b0[4];
  const size_t n = (size_t)Z * X * Y;
  const int Yh = Y / 2 + 1;
  const size_t nh = (size_t)Z * X * Yh;
  // reference spectra that are not there yet (a temporary reference: made when a crop is first needed)
  for (int c = first; c < first + count; ++c) {
    std::lock_guard<std::mutex> lk(ref->mu);
    if (ref->have[c]) continue;
    ref->spec[c] = ws_get(nh * sizeof(cplx));   // (a temporary reference lives for one call of one thread: cache blocks)
    if (!ref->spec[c]) return IA3_ENOMEM;
    ref->cached[c] = true;
    Scratch rb(n * sizeof(double));
    if (!rb.p) return IA3_ENOMEM;
    rc = crop_spectrum(ref->ref, ref->box[c], (cplx*)ref->spec[c], rb.as<double>(), st); if (rc) return rc;
    ref->have[c] = true;
  }
  const double u = (double)upsample;
  const int R = (int)ceil(u * 1.5);
  const double dftshift = trunc(R / 2.0);
  const int maxN = Y > X ? (Y > Z ? Y : Z) : (X > Z ? X : Z);
  const int NB = 512;
  Scratch rbuf(n * sizeof(double)), fb(nh * sizeof(cplx)), fp(nh * sizeof(cplx)), parts(2 * ABS2_BLOCKS * sizeof(double)),
      pv(NB * sizeof(double)), pi(NB * sizeof(long long)), rec(3 * sizeof(CoarseRec)),
      K((size_t)R * maxN * sizeof(cplx)), t1((size_t)R * Z * X * sizeof(cplx)), t2((size_t)R * R * Z * sizeof(cplx)),
      t3((size_t)R * R * R * sizeof(cplx)), tp((size_t)R * Z * X * sizeof(cplx));
  if (!rbuf.p || !fb.p || !fp.p || !parts.p || !pv.p || !pi.p || !rec.p || !K.p || !t1.p || !t2.p || !t3.p || !tp.p) return IA3_ENOMEM;
  void *mh = nullptr, *md = nullptr;
  constexpr size_t OFF = 1280;   // the seed stage uses the first 64 bytes of the mailbox, the peak record 1024.., the fit 2048 and up
  if (host_mailbox(4096, &mh, &md) != IA3_OK) return IA3_ENOMEM;
  static thread_local unsigned t_seq = 0;
  unsigned seqs[3];
  PlanLease fwd, inv;
  { int prc = get_plan(HIPFFT_D2Z, Z, X, Y, st, fwd); if (prc) return prc; }
  { int prc = get_plan(HIPFFT_Z2D, Z, X, Y, st, inv); if (prc) return prc; }
  auto matmul = [&](const cplx* Kmat, const cplx* in, cplx* out, int M, int cols, int ldk, int ldi) {
    if (g_dft_valu)
      hipLaunchKernelGGL(dft_contract_k, dim3((M + 63) / 64, (R + 15) / 16), dim3(256), 0, st, Kmat, in, out, R, M, cols, ldk, ldi);
    else
      hipLaunchKernelGGL(dft_contract_mfma_k, dim3((M + 127) / 128, (R + 31) / 32), dim3(256), 0, st, Kmat, in, out, R, M, cols, ldk, ldi);
  };
  for (int c = first; c < first + count; ++c) {
    const int* b = ref->box[c];
    CoarseRec* cr = rec.as<CoarseRec>() + (c - first);
    const unsigned nblk = (unsigned)((n + 255) / 256);
    {
    ProfScope ps_fft("xcorr_fft");   // crop -> float64, D2Z, cross-power, Z2D, coarse peak
    if (src->dtype == IA3_F32) hipLaunchKernelGGL((crop_to_real_k<float>), dim3((unsigned)((Y + 511) / 512), (unsigned)X, (unsigned)Z), dim3(256), 0, st, (const float*)src->d, src->X, src->Y, b[0], b[2], b[4], Z, X, Y, rbuf.as<double>());
    else hipLaunchKernelGGL((crop_to_real_k<uint16_t>), dim3((unsigned)((Y + 511) / 512), (unsigned)X, (unsigned)Z), dim3(256), 0, st, (const uint16_t*)src->d, src->X, src->Y, b[0], b[2], b[4], Z, X, Y, rbuf.as<double>());
    IA3_FFT(hipfftExecD2Z(fwd.h, rbuf.as<double>(), fb.as<cplx>()));
    // fp := prod, fb := conj(prod) (the data of the upsampled DFT); the power sums are not needed for the shift
    hipLaunchKernelGGL(half_power2_k, dim3(ABS2_BLOCKS), dim3(256), 0, st, (const cplx*)ref->spec[c], (const cplx*)fb.as<cplx>(), nh, Yh, Y, normalization,
                       fp.as<cplx>(), fb.as<cplx>(), parts.as<double>(), parts.as<double>() + ABS2_BLOCKS);
    IA3_FFT(hipfftExecZ2D(inv.h, fp.as<cplx>(), rbuf.as<double>()));
    hipLaunchKernelGGL(real_argmax_part_k, dim3(NB), dim3(256), 0, st, (const double*)rbuf.as<double>(), n, pv.as<double>(), pi.as<long long>());
    hipLaunchKernelGGL(coarse_final_k, dim3(1), dim3(256), 0, st, (const double*)pv.as<double>(), (const long long*)pi.as<long long>(), NB, Z, X, Y, u, dftshift, cr);
    }
    ProfScope ps_dft("xcorr_dft");   // upsampled DFT (three contractions on the f64 matrix cores) + fine peak
    // upsampled DFT around the coarse peak: axes Y (half spectrum + Hermitian completion), X, Z
    hipLaunchKernelGGL(dft_kernel_dev_k, dim3((Y + 255) / 256, R), dim3(256), 0, st, K.as<cplx>(), R, Y, (const double*)&cr->off[2], u);
    const int inner = (Y - 1) / 2;
    matmul((const cplx*)K.as<cplx>() + 1, (const cplx*)fb.as<cplx>() + 1, tp.as<cplx>(), Z * X, inner, Y, Yh);
    hipLaunchKernelGGL(half_combine_k, dim3((unsigned)(((size_t)R * Z * X + 255) / 256)), dim3(256), 0, st, (const cplx*)tp.as<cplx>(),
                       (const cplx*)fb.as<cplx>(), (const cplx*)K.as<cplx>(), Y, R, Z, X, Yh, Y, t1.as<cplx>());
    hipLaunchKernelGGL(dft_kernel_dev_k, dim3((X + 255) / 256, R), dim3(256), 0, st, K.as<cplx>(), R, X, (const double*)&cr->off[1], u);
    matmul((const cplx*)K.as<cplx>(), (const cplx*)t1.as<cplx>(), t2.as<cplx>(), R * Z, X, X, X);
    hipLaunchKernelGGL(dft_kernel_dev_k, dim3((Z + 255) / 256, R), dim3(256), 0, st, K.as<cplx>(), R, Z, (const double*)&cr->off[0], u);
    matmul((const cplx*)K.as<cplx>(), (const cplx*)t2.as<cplx>(), t3.as<cplx>(), R * R, Z, Z, Z);
    hipLaunchKernelGGL(abs_argmax_part_k, dim3(NB), dim3(256), 0, st, (const cplx*)t3.as<cplx>(), (size_t)R * R * R, pv.as<double>(), pi.as<long long>());
    seqs[c - first] = ++t_seq ? t_seq : ++t_seq;
    hipLaunchKernelGGL(fine_final_k, dim3(1), dim3(256), 0, st, (const double*)pv.as<double>(), (const long long*)pi.as<long long>(), NB, R, u, dftshift,
                       (const CoarseRec*)cr, Z, X, Y, (volatile ShiftMail*)((char*)md + OFF + (size_t)(c - first) * sizeof(ShiftMail)), seqs[c - first]);
    IA3_KCHECK();
  }
  for (int k = 0; k < count; ++k) {
    volatile ShiftMail* mb = (volatile ShiftMail*)((char*)mh + OFF + (size_t)k * sizeof(ShiftMail));
    SpinWait sw;
    while (mb->seq != seqs[k]) {
      sw.relax();
      if (((sw.n & 0xfffff) == 0 || (sw.n > 40400 && (sw.n & 0x3ff) == 0)) && hipStreamQuery(st) != hipErrorNotReady) {
        if (mb->seq == seqs[k]) break;
        IA3_HIP(hipStreamSynchronize(st));
        if (mb->seq != seqs[k]) return set_error(IA3_EHIP, "drift of crop %d did not reach the host mailbox", first + k);
        break;
      }
    }
    for (int a = 0; a < 3; ++a) shifts[3 * k + a] = mb->shift[a];
  }
  // the plans and the scratch go back now: everything queued above has finished (the last mailbox word is written by the
  // last kernel of the last crop)
  IA3_HIP(hipStreamSynchronize(st));
  return IA3_OK;
}

}  // namespace ia3k

extern "C" {

int ia3_phase_xcorr3d(const void* ref, const void* mov, int dtype, int Z, int X, int Y, int upsample,
                      int normalization, double* shift, double* err, double* phasediff) {
  ia3_stack *a = nullptr, *b = nullptr;
  int rc = ia3_stack_upload(ref, dtype, Z, X, Y, &a); if (rc) return rc;
  rc = ia3_stack_upload(mov, dtype, Z, X, Y, &b);
  if (!rc) rc = ia3_phase_xcorr3d_dev(a, b, upsample, normalization, shift, err, phasediff);
  ia3_stack_free(a); ia3_stack_free(b);
  return rc;
}

// copy the sub-box [z0,z1) x [x0,x1) x [y0,y1) of a resident stack into a new stack (drift crops)
int ia3_stack_crop(const ia3_stack* s, int z0, int z1, int x0, int x1, int y0, int y1, ia3_stack** out) {
  int rc = ensure_init(); if (rc) return rc;
  if (!s || !out) return set_error(IA3_EINVAL, "null argument");
  if (z0 < 0 || x0 < 0 || y0 < 0 || z1 > s->Z || x1 > s->X || y1 > s->Y || z0 >= z1 || x0 >= x1 || y0 >= y1)
    return set_error(IA3_EINVAL, "bad crop [%d:%d, %d:%d, %d:%d] of (%d,%d,%d)", z0, z1, x0, x1, y0, y1, s->Z, s->X, s->Y);
  rc = ia3_stack_alloc(s->dtype, z1 - z0, x1 - x0, y1 - y0, out); if (rc) return rc;
  const size_t es = esize(s->dtype);
  hipMemcpy3DParms p = {};
  p.srcPtr = make_hipPitchedPtr(s->d, (size_t)s->Y * es, s->Y, s->X);
  p.dstPtr = make_hipPitchedPtr((*out)->d, (size_t)(y1 - y0) * es, y1 - y0, x1 - x0);
  p.srcPos = make_hipPos((size_t)y0 * es, x0, z0);
  p.dstPos = make_hipPos(0, 0, 0);
  p.extent = make_hipExtent((size_t)(y1 - y0) * es, x1 - x0, z1 - z0);
  p.kind = hipMemcpyDeviceToDevice;
  hipError_t e = hipMemcpy3DAsync(&p, stream());
  if (e != hipSuccess) { ia3_stack_free(*out); *out = nullptr; return set_error(IA3_EHIP, "crop copy failed: %s", hipGetErrorString(e)); }
  return IA3_OK;
}

}  // extern "C"
