// Histogram-peak background level (reference: io_tools/load.py:642-687 find_image_background) for a whole
// resident stack and, batched, for the per-spot neighbourhoods fit_fov_image normalises with when
// normalize_local=True (spot_tools/fitting.py:246-258; crop = io_tools/crop.py:59-88 with crop_size = 2*fit_radius).
//
//   counts = np.histogram(im, bins=edges)            [e_i, e_{i+1}), last bin closed, values outside dropped
//   height = size/50; repeat: height /= 2; peaks = scipy.signal.find_peaks(counts, height) until a peak exists,
//            giving up after max_iter halvings (the result of halving max_iter+1 is computed but discarded)
//   background = centre of the bin of the highest peak (first one on ties); fallback np.nanmedian(im)
//
// Since the highest peak passes every threshold any peak passes, the loop reduces to: highest strict local
// maximum of the counts (plateaus count once, at their middle sample, ends of the array never), accepted iff
// its count >= (size/50)/2^max_iter.  Integer work throughout: results are bit-exact.
//
// local_background_k: one 256-thread block per spot; the (<= 21^3 voxel) crop is histogrammed into LDS
// (default 6552 bins = 26 KB), the peak search runs over the LDS counters.  The rare no-peak case falls back to
// an in-block 3-pass radix select of the two middle order statistics (np.nanmedian).
// whole stack: LDS-private histograms merged with global atomics, then one block for the peak.
#include "ia3_rt.h"
#include <math.h>
#include <vector>

using namespace ia3rt;

namespace ia3k {
int stack_median_all(const ia3_stack* s, float* med_all);   // corrections.hip
}

namespace {

constexpr int MAXBINS = 16000;

struct Edges {
  const double* e;   // n edges (device)
  int n;
  int uniform;       // edges are e0 + i*step: start the search at the computed bin
  double e0, step;
};

__device__ __forceinline__ int bin_of(double v, const Edges& E) {
  const int nb = E.n - 1;
  if (!(v >= E.e[0]) || !(v <= E.e[nb])) return -1;   // outside the range, or NaN
  int b;
  if (E.uniform) {
    b = (int)((v - E.e0) / E.step);
  } else {
    int lo = 0, hi = nb;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (E.e[mid] <= v) lo = mid; else hi = mid;
    }
    b = lo;
  }
  b = b < 0 ? 0 : (b > nb - 1 ? nb - 1 : b);
  while (b > 0 && v < E.e[b]) --b;                 // exact edge comparisons decide, whatever the division rounded
  while (b < nb - 1 && v >= E.e[b + 1]) ++b;
  return b;
}

// scipy.signal._peak_finding_utils._local_maxima_1d over counts[0..nb): block-wide search for the highest peak.
// key = count << 32 | ~mid  (max => highest count, then lowest index)
__device__ __forceinline__ void best_peak_block(const unsigned int* h, int nb, unsigned long long* best /* LDS, zeroed */) {
  unsigned long long mine = 0;
  for (int i = 1 + (int)threadIdx.x; i < nb - 1; i += (int)blockDim.x) {
    const unsigned c = h[i];
    if (c > h[i - 1]) {
      int j = i + 1;
      while (j < nb - 1 && h[j] == c) ++j;
      if (h[j] < c) {
        const unsigned mid = (unsigned)((i + j - 1) / 2);
        const unsigned long long key = ((unsigned long long)c << 32) | (unsigned long long)(0xffffffffu - mid);
        mine = key > mine ? key : mine;
      }
    }
  }
  if (mine) atomicMax(best, mine);
}

__device__ __forceinline__ double threshold_of(double size, int max_iter) {
  double h = size / 50.0;
  for (int k = 0; k < max_iter; ++k) h = h / 2;
  return h;
}

__device__ __forceinline__ uint32_t fkey(float v) {
  uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

struct CropBox { int z0, z1, x0, x1, y0, y1; };

template <class T>
__global__ __launch_bounds__(256) void local_background_k(const T* __restrict__ im, int Z, int X, int Y,
                                                          const float* __restrict__ centers, int n, int crop,
                                                          Edges E, int max_iter, double* __restrict__ out) {
  extern __shared__ unsigned int h[];   // nb counters (>= 2048 words allocated)
  __shared__ unsigned long long best;
  __shared__ unsigned int sel_prefix, sel_k, n_valid;
  const int i = blockIdx.x;
  if (i >= n) return;
  const int nb = E.n - 1;
  // io_tools/crop.py:81-82: [round(c - crop), round(c + crop + 1)) clipped to the image (np.round = half to even)
  CropBox B;
  {
    const double cz = (double)centers[3 * i], cx = (double)centers[3 * i + 1], cy = (double)centers[3 * i + 2];
    const double lz = fmax(rint(cz - crop), 0.0), lx = fmax(rint(cx - crop), 0.0), ly = fmax(rint(cy - crop), 0.0);
    const double rz = fmin(rint(cz + crop + 1), (double)Z), rx = fmin(rint(cx + crop + 1), (double)X),
                 ry = fmin(rint(cy + crop + 1), (double)Y);
    B.z0 = (int)lz; B.z1 = (int)rz; B.x0 = (int)lx; B.x1 = (int)rx; B.y0 = (int)ly; B.y1 = (int)ry;
  }
  const int dz = B.z1 - B.z0, dx = B.x1 - B.x0, dy = B.y1 - B.y0;
  const long long size = (dz > 0 && dx > 0 && dy > 0) ? (long long)dz * dx * dy : 0;
  for (int k = threadIdx.x; k < nb; k += 256) h[k] = 0;
  if (threadIdx.x == 0) { best = 0; n_valid = 0; }
  __syncthreads();
  for (long long v = threadIdx.x; v < size; v += 256) {
    const int y = (int)(v % dy), x = (int)((v / dy) % dx), z = (int)(v / ((long long)dy * dx));
    const double val = (double)im[((size_t)(B.z0 + z) * X + (B.x0 + x)) * Y + (B.y0 + y)];
    const int b = bin_of(val, E);
    if (b >= 0) atomicAdd(&h[b], 1u);
  }
  __syncthreads();
  best_peak_block(h, nb, &best);
  __syncthreads();
  const unsigned long long bk = best;
  const unsigned bh = (unsigned)(bk >> 32);
  if (max_iter >= 1 && bh > 0 && (double)bh >= threshold_of((double)size, max_iter)) {
    if (threadIdx.x == 0) {
      const unsigned mid = 0xffffffffu - (unsigned)(bk & 0xffffffffu);
      out[i] = (E.e[mid] + E.e[mid + 1]) / 2;
    }
    return;
  }
  // ---- np.nanmedian(crop): two middle order statistics by radix select (11 + 11 + 10 key bits) ----------------
  __syncthreads();
  for (long long v = threadIdx.x; v < size; v += 256) {
    const int y = (int)(v % dy), x = (int)((v / dy) % dx), z = (int)(v / ((long long)dy * dx));
    const float val = (float)im[((size_t)(B.z0 + z) * X + (B.x0 + x)) * Y + (B.y0 + y)];
    if (val == val) atomicAdd(&n_valid, 1u);
  }
  __syncthreads();
  const unsigned nv = n_valid;
  if (nv == 0) { if (threadIdx.x == 0) out[i] = NAN; return; }
  float two[2];
  for (int r = 0; r < 2; ++r) {
    if (threadIdx.x == 0) { sel_prefix = 0; sel_k = r == 0 ? (nv - 1) / 2 : nv / 2; }
    for (int pass = 0; pass < 3; ++pass) {
      const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
      const uint32_t dmask = pass == 2 ? 0x3ffu : 0x7ffu;
      const int hi_shift = pass == 0 ? 32 : (pass == 1 ? 21 : 10);
      __syncthreads();
      for (int k = threadIdx.x; k < 2048; k += 256) h[k] = 0;
      __syncthreads();
      const uint32_t pre = sel_prefix;
      for (long long v = threadIdx.x; v < size; v += 256) {
        const int y = (int)(v % dy), x = (int)((v / dy) % dx), z = (int)(v / ((long long)dy * dx));
        const float val = (float)im[((size_t)(B.z0 + z) * X + (B.x0 + x)) * Y + (B.y0 + y)];
        if (val == val) {
          const uint32_t key = fkey(val);
          const bool match = hi_shift >= 32 ? true : ((key >> hi_shift) == (pre >> hi_shift));
          if (match) atomicAdd(&h[(key >> shift) & dmask], 1u);
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        const int nbk = pass == 2 ? 1024 : 2048;
        unsigned k = sel_k, cum = 0;
        int b = 0;
        for (; b < nbk; ++b) { const unsigned c = h[b]; if (cum + c > k) break; cum += c; }
        if (b >= nbk) b = nbk - 1;
        sel_prefix = pre | ((uint32_t)b << shift);
        sel_k = k - cum;
      }
    }
    __syncthreads();
    two[r] = fkey_inv(sel_prefix);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[i] = (double)((two[0] + two[1]) / 2.0f);
}

template <class T>
__global__ __launch_bounds__(256) void hist_stack_k(const T* __restrict__ im, size_t n, Edges E,
                                                    unsigned int* __restrict__ hist) {
  extern __shared__ unsigned int h[];
  const int nb = E.n - 1;
  for (int k = threadIdx.x; k < nb; k += 256) h[k] = 0;
  __syncthreads();
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
    const int b = bin_of((double)im[v], E);
    if (b >= 0) atomicAdd(&h[b], 1u);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < nb; k += 256) if (h[k]) atomicAdd(&hist[k], h[k]);
}

// res[0] = background, res[1] = 1 if no peak passed (caller falls back to the median)
__global__ __launch_bounds__(256) void peak_stack_k(const unsigned int* __restrict__ hist, Edges E, double size,
                                                    int max_iter, double* __restrict__ res) {
  __shared__ unsigned long long best;
  if (threadIdx.x == 0) best = 0;
  __syncthreads();
  best_peak_block(hist, E.n - 1, &best);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long bk = best;
    const unsigned bh = (unsigned)(bk >> 32);
    if (max_iter >= 1 && bh > 0 && (double)bh >= threshold_of(size, max_iter)) {
      const unsigned mid = 0xffffffffu - (unsigned)(bk & 0xffffffffu);
      res[0] = (E.e[mid] + E.e[mid + 1]) / 2;
      res[1] = 0;
    } else {
      res[0] = NAN;
      res[1] = 1;
    }
  }
}

int make_edges(const double* edges, int n_edges, Scratch& dev, Edges& E) {
  if (!edges || n_edges < 3) return set_error(IA3_EINVAL, "need at least 3 histogram edges");
  if (n_edges - 1 > MAXBINS) return set_error(IA3_EUNSUPPORTED, "more than %d histogram bins", MAXBINS);
  for (int i = 1; i < n_edges; ++i)
    if (!(edges[i] > edges[i - 1])) return set_error(IA3_EINVAL, "histogram edges must increase");
  if (!dev.p) return IA3_ENOMEM;
  hipError_t e = hipMemcpyAsync(dev.p, edges, (size_t)n_edges * sizeof(double), hipMemcpyHostToDevice, stream());
  if (e != hipSuccess) return set_error(IA3_EHIP, "edge upload failed: %s", hipGetErrorString(e));
  E.e = dev.as<double>(); E.n = n_edges; E.e0 = edges[0]; E.step = edges[1] - edges[0];
  E.uniform = 1;
  for (int i = 0; i < n_edges; ++i)
    if (edges[i] != E.e0 + i * E.step) { E.uniform = 0; break; }
  return IA3_OK;
}

}  // namespace

extern "C" {

int ia3_find_background_dev(const ia3_stack* im, const double* edges, int n_edges, int max_iter, double* background) {
  int rc = ensure_init(); if (rc) return rc;
  if (!im || !background) return set_error(IA3_EINVAL, "null argument");
  hipStream_t st = stream();
  Scratch de((size_t)(n_edges > 0 ? n_edges : 1) * sizeof(double));
  Edges E;
  rc = make_edges(edges, n_edges, de, E); if (rc) return rc;
  const int nb = n_edges - 1;
  const size_t n = (size_t)im->Z * im->X * im->Y;
  Scratch dh((size_t)nb * sizeof(unsigned int)), dres(2 * sizeof(double));
  if (!dh.p || !dres.p) return IA3_ENOMEM;
  IA3_HIP(hipMemsetAsync(dh.p, 0, (size_t)nb * sizeof(unsigned int), st));
  const size_t lds = (size_t)(nb > 2048 ? nb : 2048) * sizeof(unsigned int);
  size_t want = (n + 256 * 64 - 1) / (256 * 64);
  unsigned blocks = (unsigned)(want < 1 ? 1 : (want > (size_t)num_cus() * 4 ? (size_t)num_cus() * 4 : want));
  {
    ProfScope ps("background_hist");
    if (im->dtype == IA3_F32)
      hipLaunchKernelGGL((hist_stack_k<float>), dim3(blocks), dim3(256), lds, st, (const float*)im->d, n, E, dh.as<unsigned int>());
    else
      hipLaunchKernelGGL((hist_stack_k<uint16_t>), dim3(blocks), dim3(256), lds, st, (const uint16_t*)im->d, n, E, dh.as<unsigned int>());
    hipLaunchKernelGGL(peak_stack_k, dim3(1), dim3(256), 0, st, (const unsigned int*)dh.p, E, (double)n, max_iter, dres.as<double>());
  }
  IA3_KCHECK();
  double res[2];
  IA3_HIP(hipMemcpyAsync(res, dres.p, sizeof(res), hipMemcpyDeviceToHost, st));
  IA3_HIP(hipStreamSynchronize(st));
  if (res[1] != 0) {   // no histogram peak: np.nanmedian(im)
    float m = 0;
    rc = ia3k::stack_median_all(im, &m); if (rc) return rc;
    res[0] = (double)m;
  }
  *background = res[0];
  return IA3_OK;
}

int ia3_local_background_dev(const ia3_stack* im, const float* centers_zxy, int n, int crop_size,
                             const double* edges, int n_edges, int max_iter, double* backgrounds) {
  int rc = ensure_init(); if (rc) return rc;
  if (!im || (n > 0 && (!centers_zxy || !backgrounds))) return set_error(IA3_EINVAL, "null argument");
  if (n <= 0) return IA3_OK;
  if (crop_size < 0 || crop_size > 64) return set_error(IA3_EINVAL, "crop_size %d not in 0..64", crop_size);
  hipStream_t st = stream();
  Scratch de((size_t)(n_edges > 0 ? n_edges : 1) * sizeof(double));
  Edges E;
  rc = make_edges(edges, n_edges, de, E); if (rc) return rc;
  const int nb = n_edges - 1;
  Scratch dc((size_t)n * 3 * sizeof(float)), dout((size_t)n * sizeof(double));
  if (!dc.p || !dout.p) return IA3_ENOMEM;
  IA3_HIP(hipMemcpyAsync(dc.p, centers_zxy, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, st));
  const size_t lds = (size_t)(nb > 2048 ? nb : 2048) * sizeof(unsigned int);
  {
    ProfScope ps("local_background");
    if (im->dtype == IA3_F32)
      hipLaunchKernelGGL((local_background_k<float>), dim3((unsigned)n), dim3(256), lds, st, (const float*)im->d, im->Z, im->X,
                         im->Y, (const float*)dc.p, n, crop_size, E, max_iter, dout.as<double>());
    else
      hipLaunchKernelGGL((local_background_k<uint16_t>), dim3((unsigned)n), dim3(256), lds, st, (const uint16_t*)im->d, im->Z,
                         im->X, im->Y, (const float*)dc.p, n, crop_size, E, max_iter, dout.as<double>());
  }
  IA3_KCHECK();
  IA3_HIP(hipMemcpyAsync(backgrounds, dout.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
  IA3_HIP(hipStreamSynchronize(st));
  return IA3_OK;
}

int ia3_find_background(const void* im, int dtype, int Z, int X, int Y, const double* edges, int n_edges,
                        int max_iter, double* background) {
  ia3_stack* s = nullptr;
  int rc = ia3_stack_upload(im, dtype, Z, X, Y, &s); if (rc) return rc;
  rc = ia3_find_background_dev(s, edges, n_edges, max_iter, background);
  ia3_stack_free(s);
  return rc;
}

}  // extern "C"
