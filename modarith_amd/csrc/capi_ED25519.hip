// modarith_amd/csrc/capi_ED25519.hip -- C-ABI entry points of the batched curve layer for ED25519 (Edwards).
#include "generated/curve_ED25519.h"
#include "edwards.h"
#define MA_CURVE_CLASS ma::Edwards<ma::C_ED25519>
#define MA_CNAME ed25519
#include "capi_curve.inc"
