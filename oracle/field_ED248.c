/* oracle/field_ED248.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * The field.c function set of `python3 monty.py 64 ED248` (5 limbs of 51 bits) for the curve-layer restatement: the
 * generic oracle bound to the constants captured from the reference (tests/golden/field_ED248.json "params"; pinned by
 * tests/test_generic_oracle.py).
 */
#include "oracle_types.h"
#define PRIME ED248
#define ORACLE_MONTGOMERY
#define NL 5
#define RADIX 51
#define NBITS 251
#define NBYTES 32
#define PM1D2 1
#include "field_bound.inc"
