/* oracle/rfc7748_oracle.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 * X25519 and X448 instances of the ladder restatement; curve constants from rfc7748.c:120-132. */
#include "oracle_types.h"

#define PRIME X25519
#define NL 5
#define NBITS 255
#define NBYTES 32
#define A24 121665
#define COF 3
#include "rfc7748_body.inc"
#undef PRIME
#undef NL
#undef NBITS
#undef NBYTES
#undef A24
#undef COF

#define PRIME X448
#define NL 8
#define NBITS 448
#define NBYTES 56
#define A24 39081
#define COF 2
#include "rfc7748_body.inc"
