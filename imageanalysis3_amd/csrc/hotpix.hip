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

extern "C" int ia3_remove_hot_pixels(const void* im, int dtype, int Z, int X, int Y, double hot_pix_th,
                                     double hot_th, void* out, int* n_hot) {
  ia3_stack* s = nullptr;
  int rc = ia3_stack_upload(im, dtype, Z, X, Y, &s); if (rc) return rc;
  hipStream_t st = stream();
  const size_t plane = (size_t)X * Y;
  std::vector<int> votes(plane);
  {
    Scratch dv(plane * sizeof(int));
    if (!dv.p) { ia3_stack_free(s); return IA3_ENOMEM; }
    dim3 g((unsigned)((Y + 63) / 64), (unsigned)((X + 3) / 4));
    if (dtype == IA3_F32) hipLaunchKernelGGL((hot_vote_k<float>), g, dim3(256), 0, st, (const float*)s->d, Z, X, Y, hot_th, dv.as<int>());
    else hipLaunchKernelGGL((hot_vote_k<uint16_t>), g, dim3(256), 0, st, (const uint16_t*)s->d, Z, X, Y, hot_th, dv.as<int>());
    hipError_t e = hipMemcpyAsync(votes.data(), dv.p, plane * sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { ia3_stack_free(s); return set_error(IA3_EHIP, "hot pixel vote failed: %s", hipGetErrorString(e)); }
  }
  std::vector<int> cand;  // np.where order: x ascending, then y
  int total = 0;
  const double th = hot_pix_th * (double)Z;
  for (int x = 0; x < X; ++x)
    for (int y = 0; y < Y; ++y)
      if ((double)votes[(size_t)x * Y + y] > th) {
        ++total;
        if (x > 0 && y > 0 && x < X - 1 && y < Y - 1) { cand.push_back(x); cand.push_back(y); }
      }
  if (n_hot) *n_hot = total;
  if (!cand.empty()) {
    Scratch dc(cand.size() * sizeof(int));
    if (!dc.p) { ia3_stack_free(s); return IA3_ENOMEM; }
    hipError_t e = hipMemcpyAsync(dc.p, cand.data(), cand.size() * sizeof(int), hipMemcpyHostToDevice, st);
    int n = (int)(cand.size() / 2);
    if (dtype == IA3_F32) hipLaunchKernelGGL((hot_fix_k<float>), dim3((Z + 63) / 64), dim3(64), 0, st, (float*)s->d, Z, X, Y, dc.as<int>(), n);
    else hipLaunchKernelGGL((hot_fix_k<uint16_t>), dim3((Z + 63) / 64), dim3(64), 0, st, (uint16_t*)s->d, Z, X, Y, dc.as<int>(), n);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { ia3_stack_free(s); return set_error(IA3_EHIP, "hot pixel fix failed: %s", hipGetErrorString(e)); }
  }
  rc = ia3_stack_download(s, out);
  ia3_stack_free(s);
  return rc;
}
