/* oracle/weierstrass_oracle.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 * NIST P-256 instance of the Weierstrass-layer restatement (weierstrass_body.inc).  Constants of curve.py:157-166
 * in internal form (value * R mod p, R = 2^260, as curve.py:244-250 converts them for a Montgomery field). */
#include "oracle_types.h"
#define CURVE nist256
#define PRIME NIST256
#define NL 5
#define NBYTES 32
#define WS_A (-3)
static const spint ws_const_b[5] = {0xdf6229c4bddfdu, 0xca8843090d89cu, 0x212ed6acf005cu, 0x83415a220abf7u, 0xc30061dd4874u};
static const spint ws_gen_x[5] = {0x30d418a9143c1u, 0xc4fedb60179e7u, 0x62251075ba95fu, 0x5c669fb732b77u, 0x8905f76b5375u};
static const spint ws_gen_y[5] = {0x5357ce95560a8u, 0x43a19e45cddf2u, 0x21f3258b4ab8eu, 0xd8552e88688ddu, 0x571ff18a5885u};
#include "weierstrass_body.inc"
