/* oracle/field_ED376.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * The field.c function set of `python3 monty.py 64 ED376` (7 limbs of 55 bits) for the curve-layer restatement: the
 * generic oracle bound to the constants captured from the reference (tests/golden/field_ED376.json "params"; pinned by
 * tests/test_generic_oracle.py).
 */
#include "oracle_types.h"
#define PRIME ED376
#define ORACLE_MONTGOMERY
#define NL 7
#define RADIX 55
#define NBITS 383
#define NBYTES 48
#define PM1D2 1
#include "field_bound.inc"
