// modarith_amd/csrc/capi_ED25519.hip -- C-ABI entry points of the batched Edwards layer for ED25519.
#include "generated/curve_ED25519.h"
#define MA_C ma::C_ED25519
#define MA_CNAME ed25519
#include "capi_edwards.inc"
