/* oracle/field_NIST521.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * The field.c function set for the Mersenne prime 2^521-1 (`python3 pseudo.py 64 NIST521`: 9 limbs of 58 bits) so
 * that the curve-layer restatement can be instantiated on it: the generic oracle bound to the constants captured
 * from the reference (tests/golden/field_NIST521.json "params"; pinned by tests/test_generic_oracle.py).
 */
#include "oracle_types.h"
#define PRIME NIST521
#define NL 9
#define RADIX 58
#define NBITS 521
#define NBYTES 66
#define PM1D2 1
#include "field_bound.inc"
