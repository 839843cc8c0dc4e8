// modarith_amd/csrc/capi_NIST256.hip -- C-ABI entry points for NIST P-256 (field only).
#include "generated/params_NIST256.h"
#define MA_P ma::P_NIST256
#define MA_NAME NIST256
#include "capi_prime.inc"
