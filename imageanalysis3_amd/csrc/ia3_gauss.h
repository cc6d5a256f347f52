// Device helpers shared by the Gaussian translation units (gauss.hip, gauss_col_*.hip).
#pragma once
#include "ia3_rt.h"
#include <type_traits>

#include "ia3_gauss_dev.h"

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
int column_kernel_source(bool f32, int Z);   // gauss_col_dispatch.hip: 0 none, 1 built in, 2 compiled at run time
int folded_axis0_f32(const float* src, int Z, size_t plane, const Taps& t, int mode, float* dst, hipStream_t s, int cert);
int folded_axis0_u16(const uint16_t* src, int Z, size_t plane, const Taps& t, int mode, uint16_t* dst, hipStream_t s, int cert);
int folded_pair_f32(const float* src, int Z, size_t plane, const Taps& bt, float* dst, const Taps& ft, float* fdst, hipStream_t s, int cert,
                    float* smin, float* sabs, int Y);
int folded_pair_u16(const uint16_t* src, int Z, size_t plane, const Taps& bt, uint16_t* dst, const Taps& ft, uint16_t* fdst, hipStream_t s,
                    int cert, float* smin, float* sabs, int Y);
}  // namespace ia3g
