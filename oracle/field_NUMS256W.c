/* oracle/field_NUMS256W.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * The field.c function set of `python3 pseudo.py 64 NUMS256W` (5 limbs of 52 bits) for the curve-layer restatement: the
 * generic oracle bound to the constants captured from the reference (tests/golden/field_NUMS256W.json "params"; pinned by
 * tests/test_generic_oracle.py).
 */
#include "oracle_types.h"
#define PRIME NUMS256W
#define NL 5
#define RADIX 52
#define NBITS 256
#define NBYTES 32
#define PM1D2 1
#include "field_bound.inc"
