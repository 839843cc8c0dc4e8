/* oracle/field_SECP256K1.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * The field.c function set of `python3 pseudo.py 64 SECP256K1` (5 limbs of 52 bits) for the curve-layer restatement: the
 * generic oracle bound to the constants captured from the reference (tests/golden/field_SECP256K1.json "params"; pinned by
 * tests/test_generic_oracle.py).
 */
#include "oracle_types.h"
#define PRIME SECP256K1
#define NL 5
#define RADIX 52
#define NBITS 256
#define NBYTES 32
#define PM1D2 1
#include "field_bound.inc"
