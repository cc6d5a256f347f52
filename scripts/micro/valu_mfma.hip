// Do the f64 vector unit and the f64 matrix cores of gfx950 run side by side?  (developer probe, not part of the library)
//   mode 0: 64 independent v_fma_f64 per loop turn            (vector unit alone)
//   mode 1: 16 independent v_mfma_f64_16x16x4_f64 per turn    (matrix cores alone)
//   mode 2: both instruction streams interleaved in the same wave
//   mode 3: half of the waves of every block run mode 0, the other half mode 1
// Reports lane-FMA/s of each stream; if the combined rate of modes 2 / 3 exceeds either stream alone the two pipelines
// overlap and a VALU-bound kernel can hand part of its multiply-adds to the matrix cores.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a, double b) {
  double x[16];
  v4d acc[4];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = (double)(threadIdx.x + i) * 1e-3;
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = v4d{0, 0, 0, 0};
  const int wv = threadIdx.x >> 6;
  const bool do_v = MODE == 0 || MODE == 2 || (MODE == 3 && (wv & 1) == 0);
  const bool do_m = MODE == 1 || MODE == 2 || (MODE == 3 && (wv & 1) == 1);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (do_m) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b + i, acc[i], 0, 0, 0);
      }
      if (do_v) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_fma(x[i], a, b);
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, int blocks, int iters) {
  double* d; hipMalloc(&d, (size_t)blocks * 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)blocks * 4;
  const double fv = (MODE == 3 ? 0.5 : (MODE == 1 ? 0 : 1)) * waves * iters * 4.0 * 16 * 64;          // lane FMAs on the vector unit
  const double fm = (MODE == 3 ? 0.5 : (MODE == 0 ? 0 : 1)) * waves * iters * 4.0 * 4 * 1024;          // multiply-adds on the matrix cores
  printf("%-28s %.2f ms   vector %.1f TFLOP/s   matrix %.1f TFLOP/s   sum %.1f\n", name, ms, 2 * fv / (ms * 1e-3) / 1e12,
         2 * fm / (ms * 1e-3) / 1e12, 2 * (fv + fm) / (ms * 1e-3) / 1e12);
  hipFree(d);
}
int main() {
  const int blocks = 256 * 8, iters = 4000;   // 8 waves per SIMD
  run<0>("vector fma alone", blocks, iters);
  run<1>("mfma f64 alone", blocks, iters);
  run<2>("interleaved in one wave", blocks, iters);
  run<3>("alternate waves", blocks, iters);
  return 0;
}
