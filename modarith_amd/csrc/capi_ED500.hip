// modarith_amd/csrc/capi_ED500.hip -- C-ABI entry points of the batched curve layer for the Edwards curve ED500 of
// curve.py:127-135 (symbols ecn_ed500_*); constants in generated/curve_ED500.h.
#include "generated/curve_ED500.h"
#include "edwards.h"
#define MA_CURVE_CLASS ma::Edwards<ma::C_ED500>
#define MA_CNAME ed500
#include "capi_curve.inc"
