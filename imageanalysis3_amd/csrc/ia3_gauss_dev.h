// Device-only helpers of the Gaussian kernels: no host headers, so that the same text compiles under hipcc (through
// ia3_gauss.h) and under hiprtc (gauss_col_dispatch.hip builds the column kernel of a stack depth that has no translation
// unit of its own at run time).  Needs IA3_MODE_NEAREST (include/ia3.h, or the run-time compiler's preamble).
#pragma once
#include <type_traits>
#include <stdint.h>

namespace ia3g {

__host__ __device__ __forceinline__ int border_idx(int q, int n, int mode) {
  if (mode == IA3_MODE_NEAREST) return q < 0 ? 0 : (q >= n ? n - 1 : q);
  if (q >= 0 && q < n) return q;
  int p = 2 * n;
  q %= p;
  if (q < 0) q += p;
  return q < n ? q : p - 1 - q;
}

template <class T> __device__ __forceinline__ double ld(const T* p, size_t i);
template <> __device__ __forceinline__ double ld<float>(const float* p, size_t i) { return (double)p[i]; }
template <> __device__ __forceinline__ double ld<uint16_t>(const uint16_t* p, size_t i) { return (double)p[i]; }
template <class T> __device__ __forceinline__ T cvt(double v);
template <> __device__ __forceinline__ float cvt<float>(double v) { return (float)v; }
template <> __device__ __forceinline__ uint16_t cvt<uint16_t>(double v) { return (uint16_t)(int)v; }

struct Taps { double w[64]; };

// ---- certified fast path of the long (VALU-bound) passes ------------------------------------------------------
// The contract fixes the f32 / u16 value of every output, not the f64 bits behind it.  For non-negative inputs the
// same sum with each (multiply, add) pair fused differs from NI_Correlate1D's by at most (2R+1) f64 ulps (all
// partial sums are non-negative and bounded by the result), so the quantised value can only differ when the fused
// sum lies within that distance of a quantisation boundary: a float32 rounding midpoint (low 29 mantissa bits
// 0x10000000) or, for uint16 truncation, an integer.  Such outputs (a few per 10^7), sums outside the normal float32
// range and threads that have seen a sign bit are recomputed with the unfused sequence; everything else takes
// two VALU instructions per tap pair instead of three.  `cert` = the guard distance in f64 ulps (4R+8 by default).
template <class T> __device__ __forceinline__ bool uncertain(double s, int cert);
template <> __device__ __forceinline__ bool uncertain<float>(double s, int cert) {
  const unsigned lo = (unsigned)__double2loint(s), hi = (unsigned)__double2hiint(s);
  // |(lo & 0x1FFFFFFF) - 0x10000000| <= cert as one unsigned range test (cert < 2^28 + ..., see cert_for and the tests' 1 << 28)
  const unsigned c = (unsigned)cert < 0x10000000u ? (unsigned)cert : 0x10000000u;
  const bool near_mid = ((lo & 0x1FFFFFFFu) - (0x10000000u - c)) <= 2u * c;
  // exponent outside [2^-100, inf): zero is exact on both paths, anything else (tiny, inf, nan) is recomputed
  const bool odd_exp = (hi - 0x39B00000u) >= (0x7FF00000u - 0x39B00000u) && (hi | lo) != 0u;
  return near_mid || odd_exp;
}
template <> __device__ __forceinline__ bool uncertain<uint16_t>(double s, int cert) {
  // |s - nearest integer| <= cert ulps of s (ulp(s) <= s * 2^-52); s == 0 is exact on both paths
  return s != 0.0 && fabs(s - rint(s)) <= s * ((double)cert * 2.220446049250313e-16);
}
template <class T> __device__ __forceinline__ unsigned sign_of(T v);
template <> __device__ __forceinline__ unsigned sign_of<float>(float v) { return __float_as_uint(v); }
template <> __device__ __forceinline__ unsigned sign_of<uint16_t>(uint16_t) { return 0u; }

// f(integral_constant<0>), f(<1>), ... while f returns true
template <int I, int N, class F>
__device__ __forceinline__ void static_for_until(F& f) {
  if constexpr (I < N) {
    if (f(std::integral_constant<int, I>{})) static_for_until<I + 1, N>(f);
  }
}

// buffer-descriptor access: vector byte offset + scalar byte offset
typedef unsigned bv4u __attribute__((ext_vector_type(4)));
typedef unsigned bv2u __attribute__((ext_vector_type(2)));
template <class T> __device__ __forceinline__ T buf_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff);
template <> __device__ __forceinline__ float buf_ld<float>(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
template <> __device__ __forceinline__ uint16_t buf_ld<uint16_t>(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0);
}
template <class T> __device__ __forceinline__ void buf_st(T v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff);
template <> __device__ __forceinline__ void buf_st<float>(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}
template <> __device__ __forceinline__ void buf_st<uint16_t>(uint16_t v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b16((short)v, r, voff, soff, 0);
}

}  // namespace ia3g
