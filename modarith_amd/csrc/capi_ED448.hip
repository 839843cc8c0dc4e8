// modarith_amd/csrc/capi_ED448.hip -- C-ABI entry points of the batched curve layer for ED448 (Edwards).
// The scalar multiplications run on the half-limb resident form of the field (csrc/fh56.h: same elements, same limbs).
#include "generated/curve_ED448.h"
#include "edwards.h"
#include "fh56.h"
#define MA_CURVE_CLASS ma::Edwards<ma::C_ED448>
#define MA_CURVE_MUL_CLASS ma::Edwards<ma::C_ED448, ma::FieldH56<ma::P_X448>>
#define MA_CNAME ed448
#include "capi_curve.inc"
