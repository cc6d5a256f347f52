// float32 instantiations of the column-in-registers axis-0 pass
#define IA3_COL_T float
#define IA3_COL_SUFFIX _f32
#include "gauss_col.inc"
