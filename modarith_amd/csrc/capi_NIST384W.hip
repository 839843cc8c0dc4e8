// modarith_amd/csrc/capi_NIST384W.hip -- C-ABI entry points of the batched curve layer for NIST P-384
// (short Weierstrass, a = -3, curve.py:168-177; symbols ecn_nist384_*, as curve.py:344-345 names them).
#include "generated/curve_NIST384.h"
#include "weierstrass.h"
#define MA_CURVE_CLASS ma::Weierstrass<ma::C_NIST384>
#define MA_CNAME nist384
#include "capi_curve.inc"
