/* oracle/edwards_ED376.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 * ED376 instance of the Edwards-layer restatement (edwards_body.inc): x^2 + y^2 = 1 -66524 x^2 y^2, CONSTANT_B and
 * CONSTANT_X kept as C ints (curve.py:117-125, 235-240), on the bound generic field oracle of its prime. */
#include "oracle_types.h"
#define CURVE ed376
#define PRIME ED376
#define NL 7
#define NBYTES 48
#define ED_A 1
#define ED_COF 2
#define ED_B_SMALL (-66524)
#define ED_SMALL_X 2
#include "edwards_body.inc"
