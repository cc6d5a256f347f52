// Sustained f64 VALU issue rate of the device under load (developer tool, not part of the library):
// independent FMA / ADD+MUL chains per lane, enough waves to fill every SIMD.  Prints instructions/s and the
// shader clock seen by s_memtime so that the Gaussian passes can be priced against what the chip sustains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE>
__global__ __launch_bounds__(256) void chains(double* out, int iters, double a, double b) {
  double x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = (double)(threadIdx.x + i) * 1e-3;
  long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) x[i] = __builtin_fma(x[i], a, b);
        else if (MODE == 1) x[i] = x[i] + a;
        else x[i] = x[i] * a;
      }
    }
  }
  long long c1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s + (double)(c1 - c0) * 0.0;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(c1 - c0);
}
template <int MODE>
void run(const char* name, int blocks, int iters) {
  double* d; hipMalloc(&d, (size_t)blocks * 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(chains<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(chains<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double cyc; hipMemcpy(&cyc, d, 8, hipMemcpyDeviceToHost);
  double instr = (double)blocks * 256 * iters * 64.0;
  printf("%s: %.2f ms, %.2f T lane-instr/s, block0 wave: %.0f counter ticks (%.3f GHz if ticks are shader clocks)\n", name, ms,
         instr / (ms * 1e-3) / 1e12, cyc, cyc / (ms * 1e-3) / 1e9);
  hipFree(d);
}
int main() {
  int blocks = 256 * 8, iters = 20000;   // 2 waves per SIMD
  run<0>("v_fma_f64", blocks, iters);
  run<1>("v_add_f64", blocks, iters);
  run<2>("v_mul_f64", blocks, iters);
  return 0;
}
