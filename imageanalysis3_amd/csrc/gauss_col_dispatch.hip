// Depth dispatch of the column-in-registers axis-0 kernels: one translation unit per (dtype, depth) is built from
// gauss_col_one.hip for every depth of IA3_FOLD_DEPTHS (Makefile: FOLD_DEPTHS, default 25 30 33 35 40 45 50 60 — the
// reference takes any `single_im_size`, io_tools/load.py:166-180; its default depth is 30, __init__.py:8-20).  Other
// depths take the sliding-window kernels (same results, +0.65 ms per 2048 x 2048 x 50 stack).
#include "ia3_gauss.h"

namespace ia3g {

#define IA3_DECL(ZZ)                                                                                                        \
  int folded_axis0_f32_z##ZZ(const float*, size_t, const Taps&, int, float*, hipStream_t, int);                             \
  int folded_axis0_u16_z##ZZ(const uint16_t*, size_t, const Taps&, int, uint16_t*, hipStream_t, int);                       \
  int folded_pair_f32_z##ZZ(const float*, size_t, const Taps&, float*, const Taps&, float*, hipStream_t, int, float*, float*, int);   \
  int folded_pair_u16_z##ZZ(const uint16_t*, size_t, const Taps&, uint16_t*, const Taps&, uint16_t*, hipStream_t, int, float*, float*, int);
IA3_FOLD_DEPTHS(IA3_DECL)
#undef IA3_DECL

int folded_axis0_f32(const float* src, int Z, size_t plane, const Taps& t, int mode, float* dst, hipStream_t s, int cert) {
  switch (Z) {
#define IA3_FOLD_CASE(ZZ) case ZZ: return folded_axis0_f32_z##ZZ(src, plane, t, mode, dst, s, cert);
    IA3_FOLD_DEPTHS(IA3_FOLD_CASE)
#undef IA3_FOLD_CASE
    default: return FOLD_NOT_COVERED;
  }
}
int folded_axis0_u16(const uint16_t* src, int Z, size_t plane, const Taps& t, int mode, uint16_t* dst, hipStream_t s, int cert) {
  switch (Z) {
#define IA3_FOLD_CASE(ZZ) case ZZ: return folded_axis0_u16_z##ZZ(src, plane, t, mode, dst, s, cert);
    IA3_FOLD_DEPTHS(IA3_FOLD_CASE)
#undef IA3_FOLD_CASE
    default: return FOLD_NOT_COVERED;
  }
}
int folded_pair_f32(const float* src, int Z, size_t plane, const Taps& bt, float* dst, const Taps& ft, float* fdst, hipStream_t s, int cert,
                    float* smin, float* sabs, int Y) {
  switch (Z) {
#define IA3_FOLD_CASE(ZZ) case ZZ: return folded_pair_f32_z##ZZ(src, plane, bt, dst, ft, fdst, s, cert, smin, sabs, Y);
    IA3_FOLD_DEPTHS(IA3_FOLD_CASE)
#undef IA3_FOLD_CASE
    default: return FOLD_NOT_COVERED;
  }
}
int folded_pair_u16(const uint16_t* src, int Z, size_t plane, const Taps& bt, uint16_t* dst, const Taps& ft, uint16_t* fdst, hipStream_t s,
                    int cert, float* smin, float* sabs, int Y) {
  switch (Z) {
#define IA3_FOLD_CASE(ZZ) case ZZ: return folded_pair_u16_z##ZZ(src, plane, bt, dst, ft, fdst, s, cert, smin, sabs, Y);
    IA3_FOLD_DEPTHS(IA3_FOLD_CASE)
#undef IA3_FOLD_CASE
    default: return FOLD_NOT_COVERED;
  }
}

}  // namespace ia3g
