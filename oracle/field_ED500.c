/* oracle/field_ED500.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * The field.c function set of `python3 monty.py 64 ED500` (9 limbs of 57 bits) for the curve-layer restatement: the
 * generic oracle bound to the constants captured from the reference (tests/golden/field_ED500.json "params"; pinned by
 * tests/test_generic_oracle.py).
 */
#include "oracle_types.h"
#define PRIME ED500
#define ORACLE_MONTGOMERY
#define NL 9
#define RADIX 57
#define NBITS 505
#define NBYTES 64
#define PM1D2 1
#include "field_bound.inc"
