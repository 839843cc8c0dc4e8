// modarith_amd/csrc/capi_ED448.hip -- C-ABI entry points of the batched curve layer for ED448 (Edwards).
#include "generated/curve_ED448.h"
#include "edwards.h"
#define MA_CURVE_CLASS ma::Edwards<ma::C_ED448>
#define MA_CNAME ed448
#include "capi_curve.inc"
