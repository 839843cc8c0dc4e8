// modarith_amd/csrc/capi_ED376.hip -- C-ABI entry points of the batched curve layer for the Edwards curve ED376 of
// curve.py:117-125 (symbols ecn_ed376_*); constants in generated/curve_ED376.h.
#include "generated/curve_ED376.h"
#include "edwards.h"
#define MA_CURVE_CLASS ma::Edwards<ma::C_ED376>
#define MA_CNAME ed376
#include "capi_curve.inc"
