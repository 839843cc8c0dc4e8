/* oracle/weierstrass_NUMS256W.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 * NUMS256W instance of the Weierstrass-layer restatement: a = -3, CONSTANT_B = 152961, CONSTANT_X = 2
 * (curve.py:147-155, 235-240). */
#include "oracle_types.h"
#define CURVE nums256w
#define PRIME NUMS256W
#define NL 5
#define NBYTES 32
#define WS_A (-3)
#define WS_SMALL_B 152961
#define WS_SMALL_X 2
#include "weierstrass_body.inc"
