// modarith_amd/csrc/capi_NIST521W.hip -- C-ABI entry points of the batched curve layer for NIST P-521
// (short Weierstrass, a = -3, over the Mersenne prime 2^521-1, curve.py:179-188; symbols ecn_nist521_*).
#include "generated/curve_NIST521.h"
#include "weierstrass.h"
#define MA_CURVE_CLASS ma::Weierstrass<ma::C_NIST521>
#define MA_CNAME nist521
#include "capi_curve.inc"
