/* oracle/edwards_NUMS256E.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 * NUMS256E instance of the Edwards-layer restatement (edwards_body.inc): x^2 + y^2 = 1 - 15342 x^2 y^2 over 2^256-189,
 * CONSTANT_B and CONSTANT_X kept as C ints (curve.py:137-145, 235-240), on the bound generic field oracle of the
 * NUMS256W prime (the same modulus). */
#include "oracle_types.h"
#define CURVE nums256e
#define PRIME NUMS256W
#define NL 5
#define NBYTES 32
#define ED_A 1
#define ED_COF 2
#define ED_B_SMALL (-15342)
#define ED_SMALL_X 34
#include "edwards_body.inc"
