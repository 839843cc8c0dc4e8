// modarith_amd/csrc/capi_ED448.hip -- C-ABI entry points of the batched Edwards layer for ED448.
#include "generated/curve_ED448.h"
#define MA_C ma::C_ED448
#define MA_CNAME ed448
#include "capi_edwards.inc"
