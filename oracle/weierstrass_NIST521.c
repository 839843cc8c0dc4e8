/* oracle/weierstrass_NIST521.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 * NIST P-521 instance of the Weierstrass-layer restatement (weierstrass_body.inc).  Constants of curve.py:179-188 as
 * plain limbs (pseudo-Mersenne field: no Montgomery conversion, curve.py:244-250). */
#include "oracle_types.h"
#define CURVE nist521
#define PRIME NIST521
#define NL 9
#define NBYTES 66
#define WS_A (-3)
static const spint ws_const_b[9] = {0x3451fd46b503f00u, 0xf7e20f4b0d3c7bu, 0xbd3bb1bf07357u, 0x147b1fa4dec594bu, 0x18ef109e1561939u, 0x26cc57cee2d2264u, 0x540eea2da725b9u, 0x2687e4a688682dau, 0x51953eb9618e1cu};
static const spint ws_gen_x[9] = {0x17e7e31c2e5bd66u, 0x22cf0615a90a6feu, 0x127a2ffa8de334u, 0x1dfbf9d64a3f877u, 0x6b4d3dbaa14b5eu, 0x14fed487e0a2bd8u, 0x15b4429c6481390u, 0x3a73678fb2d988eu, 0xc6858e06b70404u};
static const spint ws_gen_y[9] = {0xbe94769fd16650u, 0x31c21a89cb09022u, 0x39013fad0761353u, 0x2657bd099031542u, 0x3273e662c97ee72u, 0x1e6d11a05ebef45u, 0x3d1bd998f544495u, 0x3001172297ed0b1u, 0x11839296a789a3bu};
#include "weierstrass_body.inc"
