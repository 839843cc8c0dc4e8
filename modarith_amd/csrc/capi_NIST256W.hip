// modarith_amd/csrc/capi_NIST256W.hip -- C-ABI entry points of the batched curve layer for NIST P-256
// (short Weierstrass, a = -3; symbols ecn_nist256_*, as curve.py:344-345 names them).
#define MA_MUL_WPS 3
#include "generated/curve_NIST256.h"
#include "weierstrass.h"
#define MA_CURVE_CLASS ma::Weierstrass<ma::C_NIST256>
#define MA_CNAME nist256
#include "capi_curve.inc"
