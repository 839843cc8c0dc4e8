// modarith_amd/csrc/capi_NIST256W.hip -- C-ABI entry points of the batched curve layer for NIST P-256
// (short Weierstrass, a = -3; symbols ecn_nist256_*, as curve.py:344-345 names them).
// Three waves per SIMD: the scalar multiplications take 147-157 VGPRs since the Montgomery digits enter the columns as multiply-adds
// (field.h monty_mul_half).  A resident half-limb form of this field (as fh51.h / fh56.h) was built and measured in round 4: 128-135
// VGPRs, four waves, limbs equal -- and 3.14 against 3.17e7 ecn mul/s: the kernel is issue- and clock-bound, not occupancy-bound; not kept.
#define MA_MUL_WPS 3
#include "generated/curve_NIST256.h"
#include "weierstrass.h"
#define MA_CURVE_CLASS ma::Weierstrass<ma::C_NIST256>
#define MA_CNAME nist256
#include "capi_curve.inc"
