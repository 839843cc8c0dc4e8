// modarith_amd/csrc/capi_X25519.hip -- C-ABI entry points for X25519 (field + RFC 7748 ladder;
// curve constants from rfc7748.c:120-132).
#include "generated/params_X25519.h"
#define MA_P ma::P_X25519
#define MA_NAME X25519
#define MA_LADDER_A24 121665
#define MA_LADDER_COF 3
#define MA_LADDER_FE26 1
#include "capi_prime.inc"
