/* oracle/edwards_ED500.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 * ED500 instance of the Edwards-layer restatement (edwards_body.inc): x^2 + y^2 = 1 -105355 x^2 y^2, CONSTANT_B and
 * CONSTANT_X kept as C ints (curve.py:127-135, 235-240), on the bound generic field oracle of its prime. */
#include "oracle_types.h"
#define CURVE ed500
#define PRIME ED500
#define NL 9
#define NBYTES 64
#define ED_A 1
#define ED_COF 2
#define ED_B_SMALL (-105355)
#define ED_SMALL_X 6
#include "edwards_body.inc"
