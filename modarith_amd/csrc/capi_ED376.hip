// modarith_amd/csrc/capi_ED376.hip -- C-ABI entry points of the batched curve layer for the Edwards curve ED376 of
// curve.py:117-125 (symbols ecn_ed376_*); constants in generated/curve_ED376.h.
// The prime-limb constants of the Montgomery reduction stay literals here (field.h Wide::prep_const): as opaque scalars -- a gain for
// P-256 and P-384 -- they push k_ed_mul of this 7 x 55-bit field from 256 VGPRs to 256 + 6 spilled.
#define MA_MHALF_SHIFT_TERMS
#include "generated/curve_ED376.h"
#include "edwards.h"
#define MA_CURVE_CLASS ma::Edwards<ma::C_ED376>
#define MA_CNAME ed376
#include "capi_curve.inc"
