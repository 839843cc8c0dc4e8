// modarith_amd/csrc/capi_NUMS256WW.hip -- C-ABI entry points of the batched curve layer for the short-Weierstrass curve
// NUMS256W of curve.py (symbols ecn_nums256w_*); constants in generated/curve_NUMS256W.h.
#include "generated/curve_NUMS256W.h"
#include "weierstrass.h"
#define MA_CURVE_CLASS ma::Weierstrass<ma::C_NUMS256W>
#define MA_CNAME nums256w
#include "capi_curve.inc"
