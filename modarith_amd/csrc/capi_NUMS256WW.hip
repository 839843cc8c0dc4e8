// modarith_amd/csrc/capi_NUMS256WW.hip -- C-ABI entry points of the batched curve layer for the short-Weierstrass curve
// NUMS256W of curve.py (symbols ecn_nums256w_*); constants in generated/curve_NUMS256W.h.
// three waves per SIMD: the scalar multiplications of this curve need 130-153 VGPRs (csrc/curve.h MA_MUL_WPS; +2-5 % over two waves)
#define MA_MUL_WPS 3
#include "generated/curve_NUMS256W.h"
#include "weierstrass.h"
#define MA_CURVE_CLASS ma::Weierstrass<ma::C_NUMS256W>
#define MA_CNAME nums256w
#include "capi_curve.inc"
