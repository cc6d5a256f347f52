// One (dtype, depth) instantiation of the column-in-registers axis-0 kernels: -DIA3_COL_F32=1|0 -DIA3_COL_Z=<depth>
#include <stdint.h>
#if IA3_COL_F32
#define IA3_COL_T float
#define IA3_COL_SUFFIX _f32
#else
#define IA3_COL_T uint16_t
#define IA3_COL_SUFFIX _u16
#endif
#include "gauss_col.inc"
