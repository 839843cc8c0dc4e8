/* oracle/field_NIST384.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * The field.c function set for NIST P-384 (`python3 monty.py 64 NIST384`: 7 limbs of 56 bits, R = 2^392, full
 * Montgomery digits) so that the curve-layer restatement (weierstrass_body.inc) can be instantiated on it.  The six
 * prime-specific functions are the run-time generic oracle (field_generic.c) bound to the parameter block that
 * tests fill from the constants captured from the reference (tests/golden/field_NIST384.json "params", against
 * whose vectors the generic oracle is pinned by tests/test_generic_oracle.py); the rest is field_common.inc.
 * oracle_bind_NIST384() must be called once before anything else here.
 */
#include "oracle_types.h"
#include "field_generic.h"
#define PRIME NIST384
#define ORACLE_MONTGOMERY
#define NL 7
#define RADIX 56
#define NBITS 384
#define NBYTES 48
#define PM1D2 1

#include "field_bound.inc"
