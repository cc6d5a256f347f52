// Device helpers shared by the Gaussian translation units (gauss.hip, gauss_col_*.hip).
#pragma once
#include "ia3_rt.h"
#include <type_traits>

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

// stack depths the column kernel is instantiated for (the column lives in registers: one kernel per depth, dtype and
// variant; one translation unit per (dtype, depth), built in parallel).  The Makefile passes the list (FOLD_DEPTHS); this is
// its default for a compile outside make.
#ifndef IA3_FOLD_DEPTHS
#define IA3_FOLD_DEPTHS(X) X(25) X(30) X(33) X(35) X(40) X(45) X(50) X(60)
#endif
namespace ia3g {
inline bool fold_depth(int Z) {
  switch (Z) {
#define IA3_FOLD_CASE(ZZ) case ZZ: return true;
    IA3_FOLD_DEPTHS(IA3_FOLD_CASE)
#undef IA3_FOLD_CASE
    default: return false;
  }
}


// entry points of gauss_col_dispatch.hip (which picks the translation unit of the depth).  Return 0, a (negative) IA3 error code, or FOLD_NOT_COVERED when the depth is not
// instantiated.  IA3 error codes are negative, the two private "nothing wrong, take the other path" codes positive.
constexpr int FOLD_NOT_COVERED = 1;   // this depth / these radii have no column kernel: nothing was queued
constexpr int FOLD_NO_FORK = 2;       // dog_pair_t: no auxiliary stream, everything ran on the main one (nothing to join)
int folded_axis0_f32(const float* src, int Z, size_t plane, const Taps& t, int mode, float* dst, hipStream_t s, int cert);
int folded_axis0_u16(const uint16_t* src, int Z, size_t plane, const Taps& t, int mode, uint16_t* dst, hipStream_t s, int cert);
int folded_pair_f32(const float* src, int Z, size_t plane, const Taps& bt, float* dst, const Taps& ft, float* fdst, hipStream_t s, int cert,
                    float* smin, float* sabs, int Y);
int folded_pair_u16(const uint16_t* src, int Z, size_t plane, const Taps& bt, uint16_t* dst, const Taps& ft, uint16_t* fdst, hipStream_t s,
                    int cert, float* smin, float* sabs, int Y);
}  // namespace ia3g
