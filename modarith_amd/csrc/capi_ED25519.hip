// modarith_amd/csrc/capi_ED25519.hip -- C-ABI entry points of the batched curve layer for ED25519 (Edwards).
// The scalar multiplications run on the half-limb resident form of the field (csrc/fh51.h: same elements, same limbs, 124 VGPRs)
// at four waves per SIMD.
#define MA_MUL_WPS 4
#include "generated/curve_ED25519.h"
#include "edwards.h"
#include "fh51.h"
#define MA_CURVE_CLASS ma::Edwards<ma::C_ED25519>
#define MA_CURVE_MUL_CLASS ma::Edwards<ma::C_ED25519, ma::FieldH51<ma::P_X25519>>
#define MA_CNAME ed25519
#include "capi_curve.inc"
