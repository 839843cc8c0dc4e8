// modarith_amd/csrc/capi_X448.hip -- C-ABI entry points for X448 (field + RFC 7748 ladder;
// curve constants from rfc7748.c:120-132).
#include "generated/params_X448.h"
#define MA_P ma::P_X448
#define MA_NAME X448
#define MA_LADDER_A24 39081
#define MA_LADDER_COF 2
#define MA_LADDER_FE28 1
#include "capi_prime.inc"
