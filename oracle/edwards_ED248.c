/* oracle/edwards_ED248.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 * ED248 instance of the Edwards-layer restatement (edwards_body.inc): x^2 + y^2 = 1 -107431 x^2 y^2, CONSTANT_B and
 * CONSTANT_X kept as C ints (curve.py:107-115, 235-240), on the bound generic field oracle of its prime. */
#include "oracle_types.h"
#define CURVE ed248
#define PRIME ED248
#define NL 5
#define NBYTES 32
#define ED_A 1
#define ED_COF 2
#define ED_B_SMALL (-107431)
#define ED_SMALL_X 4
#include "edwards_body.inc"
