// Hot-pixel removal (reference: correction_tools/filter.py:22-42, twin corrections.py:490-510).
//
//   conv  = (roll(im,1,x) + roll(im,-1,x) + roll(im,1,y) + roll(im,1,y)) / 4     (the duplicated
//           roll(im,1,2) is the reference's; sums are in the stack dtype: uint16 wraps)
//   hot2d = sum_z (im > hot_th * conv);   candidates: hot2d > hot_pix_th * Z
//   interior candidates, in np.where order: column <- mean of its 4 neighbours of the image being
//   corrected (later candidates see earlier replacements).
// HBM-bound vote pass (one read of the stack); the replacement touches only the candidate columns:
// one thread per z-plane walks the candidate list in order, which preserves the reference's
// sequential dependence between adjacent hot columns without any synchronisation.
#include "ia3_rt.h"
#include <algorithm>
#include <math.h>
#include <vector>

namespace {

template <class T> struct Arith;
template <> struct Arith<uint16_t> {
  static __device__ __forceinline__ bool hot(uint16_t v, uint16_t a, uint16_t b, uint16_t c, double hot_th) {
    uint16_t s = (uint16_t)((uint16_t)((uint16_t)(a + b) + c) + c);  // uint16 wrap-around as NumPy
    return (double)v > hot_th * ((double)s / 4.0);
  }
  static __device__ __forceinline__ uint16_t mean4(uint16_t a, uint16_t b, uint16_t c, uint16_t d) {
    uint16_t s = (uint16_t)((uint16_t)((uint16_t)(a + b) + c) + d);
    return (uint16_t)(int)((double)s / 4.0);
  }
};
template <> struct Arith<float> {
  static __device__ __forceinline__ bool hot(float v, float a, float b, float c, double hot_th) {
    float conv = (((a + b) + c) + c) / 4.0f;
    return v > (float)hot_th * conv;
  }
  static __device__ __forceinline__ float mean4(float a, float b, float c, float d) {
    return (((a + b) + c) + d) / 4.0f;
  }
};

template <class T>
__global__ __launch_bounds__(256) void hot_vote_k(const T* __restrict__ im, int Z, int X, int Y, double hot_th,
                                                  int* __restrict__ votes) {
  const int y = blockIdx.x * 64 + (threadIdx.x & 63);
  const int x = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= X || y >= Y) return;
  const int xm = x == 0 ? X - 1 : x - 1, xp = x == X - 1 ? 0 : x + 1, ym = y == 0 ? Y - 1 : y - 1;
  int cnt = 0;
  for (int z = 0; z < Z; ++z) {
    const T* pl = im + (size_t)z * X * Y;
    const T v = pl[(size_t)x * Y + y];
    // roll(im,1,1)[x] = im[x-1]; roll(im,-1,1)[x] = im[x+1]; roll(im,1,2)[y] = im[y-1]
    cnt += Arith<T>::hot(v, pl[(size_t)xm * Y + y], pl[(size_t)xp * Y + y], pl[(size_t)x * Y + ym], hot_th);
  }
  votes[(size_t)x * Y + y] = cnt;
}

template <class T>
__global__ void hot_fix_k(T* __restrict__ im, int Z, int X, int Y, const int* __restrict__ cand, int n) {
  const int z = blockIdx.x * blockDim.x + threadIdx.x;
  if (z >= Z) return;
  T* pl = im + (size_t)z * X * Y;
  for (int k = 0; k < n; ++k) {
    const int x = cand[2 * k], y = cand[2 * k + 1];
    pl[(size_t)x * Y + y] = Arith<T>::mean4(pl[(size_t)(x + 1) * Y + y], pl[(size_t)(x - 1) * Y + y],
                                            pl[(size_t)x * Y + y + 1], pl[(size_t)x * Y + y - 1]);
  }
}

}  // namespace

using namespace ia3rt;

namespace {

// uint16 storage, float32 arithmetic: the chain of io_tools/load.py:323-334 calls
// corrections.Remove_Hot_Pixels(im.astype(np.float32), dtype=np.uint16, ...): votes and replacement means are float32,
// later candidates see the UNROUNDED float32 replacements of earlier ones, and the cast to uint16 (truncation)
// happens once at the end.  idx_of[x*Y+y] = candidate rank of an interior hot column (else -1); repl[k*Z+z] keeps
// the float32 replacement of candidate k in plane z.
__global__ __launch_bounds__(256) void hot_vote_u16f_k(const uint16_t* __restrict__ im, int Z, int X, int Y, double hot_th,
                                                       int* __restrict__ votes) {
  const int y = blockIdx.x * 64 + (threadIdx.x & 63);
  const int x = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= X || y >= Y) return;
  const int xm = x == 0 ? X - 1 : x - 1, xp = x == X - 1 ? 0 : x + 1, ym = y == 0 ? Y - 1 : y - 1;
  int cnt = 0;
  for (int z = 0; z < Z; ++z) {
    const uint16_t* pl = im + (size_t)z * X * Y;
    cnt += Arith<float>::hot((float)pl[(size_t)x * Y + y], (float)pl[(size_t)xm * Y + y], (float)pl[(size_t)xp * Y + y],
                             (float)pl[(size_t)x * Y + ym], hot_th);
  }
  votes[(size_t)x * Y + y] = cnt;
}

__global__ void hot_mark_k(int* __restrict__ idx_of, const int* __restrict__ cand, int n, int Y) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < n) idx_of[(size_t)cand[2 * k] * Y + cand[2 * k + 1]] = k;
}

__global__ void hot_fix_u16f_k(uint16_t* __restrict__ im, int Z, int X, int Y, const int* __restrict__ cand, int n,
                               const int* __restrict__ idx_of, float* __restrict__ repl) {
  const int z = blockIdx.x * blockDim.x + threadIdx.x;
  if (z >= Z) return;
  uint16_t* pl = im + (size_t)z * X * Y;
  auto val = [&](int x, int y, int k) -> float {
    const int j = idx_of[(size_t)x * Y + y];
    return (j >= 0 && j < k) ? repl[(size_t)j * Z + z] : (float)pl[(size_t)x * Y + y];
  };
  for (int k = 0; k < n; ++k) {
    const int x = cand[2 * k], y = cand[2 * k + 1];
    repl[(size_t)k * Z + z] = Arith<float>::mean4(val(x + 1, y, k), val(x - 1, y, k), val(x, y + 1, k), val(x, y - 1, k));
  }
  for (int k = 0; k < n; ++k) {   // _nim.astype(np.uint16)
    const float v = repl[(size_t)k * Z + z];
    pl[(size_t)cand[2 * k] * Y + cand[2 * k + 1]] = (fabsf(v) < 2147483648.f) ? (uint16_t)(int)v : (uint16_t)0;
  }
}

// Columns with more than `th` votes, compacted on the device: the host used to download the whole vote plane (16.8 MB
// for a 2048 x 2048 image) and scan it — 3-5 ms of host time per channel, four channels per movie.  ctl[0] = all hot
// columns (border ones included: n_hot), ctl[1] = entries written to `list` (interior ones, flat index x * Y + y, in
// no particular order; the host sorts them into np.where order).
constexpr int HOT_CAP = 16384;
__global__ __launch_bounds__(256) void hot_compact_k(const int* __restrict__ votes, int X, int Y, double th,
                                                     int* __restrict__ ctl, int* __restrict__ list) {
  const size_t n = (size_t)X * Y;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (!((double)votes[i] > th)) continue;
    atomicAdd(&ctl[0], 1);
    const int x = (int)(i / (size_t)Y), y = (int)(i % (size_t)Y);
    if (x > 0 && y > 0 && x < X - 1 && y < Y - 1) {
      const int k = atomicAdd(&ctl[1], 1);
      if (k < HOT_CAP) list[k] = (int)i;
    }
  }
}

int hot_pixels_inplace(ia3_stack* s, double hot_pix_th, double hot_th, int float_arith, int* n_hot) {
  hipStream_t st = stream();
  const int dtype = s->dtype, Z = s->Z, X = s->X, Y = s->Y;
  const size_t plane = (size_t)X * Y;
  if (plane > 0x7fffffffull) return set_error(IA3_EUNSUPPORTED, "hot pixels: plane of %d x %d exceeds 2^31 pixels", X, Y);
  if (float_arith && dtype != IA3_U16) float_arith = 0;   // a float32 stack already computes in float32
  Scratch dv(plane * sizeof(int)), dl((size_t)(2 + HOT_CAP) * sizeof(int));
  if (!dv.p || !dl.p) return IA3_ENOMEM;
  hipError_t e = hipMemsetAsync(dl.p, 0, 2 * sizeof(int), st);
  if (e != hipSuccess) return set_error(IA3_EHIP, "memset failed: %s", hipGetErrorString(e));
  {
    ProfScope ps("hot_vote");
    dim3 g((unsigned)((Y + 63) / 64), (unsigned)((X + 3) / 4));
    if (dtype == IA3_F32) hipLaunchKernelGGL((hot_vote_k<float>), g, dim3(256), 0, st, (const float*)s->d, Z, X, Y, hot_th, dv.as<int>());
    else if (float_arith) hipLaunchKernelGGL(hot_vote_u16f_k, g, dim3(256), 0, st, (const uint16_t*)s->d, Z, X, Y, hot_th, dv.as<int>());
    else hipLaunchKernelGGL((hot_vote_k<uint16_t>), g, dim3(256), 0, st, (const uint16_t*)s->d, Z, X, Y, hot_th, dv.as<int>());
    const double th = hot_pix_th * (double)Z;
    hipLaunchKernelGGL(hot_compact_k, dim3(1024), dim3(256), 0, st, (const int*)dv.as<int>(), X, Y, th, dl.as<int>(), dl.as<int>() + 2);
  }
  IA3_KCHECK();
  // the two counts, then the (few) entries: pinned mailbox words would save ~20 us more; this runs once per channel
  int ctl[2] = {0, 0};
  e = hipMemcpyAsync(ctl, dl.p, sizeof(ctl), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return set_error(IA3_EHIP, "hot pixel vote failed: %s", hipGetErrorString(e));
  // np.where order: x ascending, then y.  (thread-local: the upload below reads it after this function has returned;
  // the next call on this thread synchronises the stream before it touches the vector again)
  static thread_local std::vector<int> cand;
  cand.clear();
  int total = ctl[0];
  if (ctl[1] > HOT_CAP) {   // more interior hot columns than the device list holds: the whole vote plane, scanned here
    std::vector<int> votes(plane);
    e = hipMemcpyAsync(votes.data(), dv.p, plane * sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return set_error(IA3_EHIP, "hot pixel vote download failed: %s", hipGetErrorString(e));
    const double th = hot_pix_th * (double)Z;
    total = 0;
    for (int x = 0; x < X; ++x)
      for (int y = 0; y < Y; ++y)
        if ((double)votes[(size_t)x * Y + y] > th) {
          ++total;
          if (x > 0 && y > 0 && x < X - 1 && y < Y - 1) { cand.push_back(x); cand.push_back(y); }
        }
  } else if (ctl[1] > 0) {
    std::vector<int> flat((size_t)ctl[1]);
    e = hipMemcpyAsync(flat.data(), dl.as<int>() + 2, flat.size() * sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return set_error(IA3_EHIP, "hot pixel list download failed: %s", hipGetErrorString(e));
    std::sort(flat.begin(), flat.end());
    cand.reserve(flat.size() * 2);
    for (int i : flat) { cand.push_back(i / Y); cand.push_back(i % Y); }
  }
  if (n_hot) *n_hot = total;
  if (cand.empty()) return IA3_OK;
  const int n = (int)(cand.size() / 2);
  Scratch dc(cand.size() * sizeof(int));
  if (!dc.p) return IA3_ENOMEM;
  e = hipMemcpyAsync(dc.p, cand.data(), cand.size() * sizeof(int), hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return set_error(IA3_EHIP, "hot pixel list upload failed: %s", hipGetErrorString(e));
  ProfScope ps("hot_fix");
  if (dtype == IA3_F32) {
    hipLaunchKernelGGL((hot_fix_k<float>), dim3((Z + 63) / 64), dim3(64), 0, st, (float*)s->d, Z, X, Y, dc.as<int>(), n);
  } else if (!float_arith) {
    hipLaunchKernelGGL((hot_fix_k<uint16_t>), dim3((Z + 63) / 64), dim3(64), 0, st, (uint16_t*)s->d, Z, X, Y, dc.as<int>(), n);
  } else {
    Scratch dr((size_t)n * Z * sizeof(float));
    if (!dr.p) return IA3_ENOMEM;
    e = hipMemsetAsync(dv.p, 0xFF, plane * sizeof(int), st);   // -1 everywhere
    if (e != hipSuccess) return set_error(IA3_EHIP, "memset failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(hot_mark_k, dim3((n + 255) / 256), dim3(256), 0, st, dv.as<int>(), dc.as<int>(), n, Y);
    hipLaunchKernelGGL(hot_fix_u16f_k, dim3((Z + 63) / 64), dim3(64), 0, st, (uint16_t*)s->d, Z, X, Y, dc.as<int>(), n,
                       (const int*)dv.p, dr.as<float>());
    // (dr / dv go back to the scratch cache when this scope ends; the cache orders their reuse behind this stream)
  }
  IA3_KCHECK();
  return IA3_OK;
}

}  // namespace

extern "C" int ia3_remove_hot_pixels_dev(ia3_stack* im, double hot_pix_th, double hot_th, int float_arith, int* n_hot) {
  int rc = ensure_init(); if (rc) return rc;
  if (!im) return set_error(IA3_EINVAL, "null stack");
  return hot_pixels_inplace(im, hot_pix_th, hot_th, float_arith, n_hot);
}

extern "C" int ia3_remove_hot_pixels(const void* im, int dtype, int Z, int X, int Y, double hot_pix_th,
                                     double hot_th, void* out, int* n_hot) {
  ia3_stack* s = nullptr;
  int rc = ia3_stack_upload(im, dtype, Z, X, Y, &s); if (rc) return rc;
  rc = hot_pixels_inplace(s, hot_pix_th, hot_th, 0, n_hot);
  if (rc == IA3_OK) rc = ia3_stack_download(s, out);
  ia3_stack_free(s);
  return rc;
}
