// modarith_amd/csrc/capi_SECP256K1W.hip -- C-ABI entry points of the batched curve layer for the short-Weierstrass curve
// SECP256K1 of curve.py (symbols ecn_secp256k1_*); constants in generated/curve_SECP256K1.h.
// three waves per SIMD: the scalar multiplications of this curve need 130-153 VGPRs (csrc/curve.h MA_MUL_WPS; +2-5 % over two waves)
#define MA_MUL_WPS 3
#include "generated/curve_SECP256K1.h"
#include "weierstrass.h"
#define MA_CURVE_CLASS ma::Weierstrass<ma::C_SECP256K1>
#define MA_CNAME secp256k1
#include "capi_curve.inc"
