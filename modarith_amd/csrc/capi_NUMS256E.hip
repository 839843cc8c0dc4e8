// modarith_amd/csrc/capi_NUMS256E.hip -- C-ABI entry points of the batched curve layer for the Edwards curve NUMS256E
// of curve.py:137-145 (x^2 + y^2 = 1 - 15342 x^2 y^2 over 2^256-189; symbols ecn_nums256e_*).
// three waves per SIMD: the scalar multiplications of this curve need 130-153 VGPRs (csrc/curve.h MA_MUL_WPS; +2-5 % over two waves)
#define MA_MUL_WPS 3
#include "generated/curve_NUMS256E.h"
#include "edwards.h"
#define MA_CURVE_CLASS ma::Edwards<ma::C_NUMS256E>
#define MA_CNAME nums256e
#include "capi_curve.inc"
