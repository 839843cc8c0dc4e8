// modarith_amd/csrc/capi_ED248.hip -- C-ABI entry points of the batched curve layer for the Edwards curve ED248 of
// curve.py:107-115 (symbols ecn_ed248_*); constants in generated/curve_ED248.h.
#include "generated/curve_ED248.h"
#include "edwards.h"
#define MA_CURVE_CLASS ma::Edwards<ma::C_ED248>
#define MA_CNAME ed248
#include "capi_curve.inc"
