// uint16 instantiations of the column-in-registers axis-0 pass
#include <stdint.h>
#define IA3_COL_T uint16_t
#define IA3_COL_SUFFIX _u16
#include "gauss_col.inc"
