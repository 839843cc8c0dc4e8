/* oracle/weierstrass_NIST384.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 * NIST P-384 instance of the Weierstrass-layer restatement (weierstrass_body.inc).  Constants of curve.py:168-177
 * in internal form (value * R mod p, R = 2^392, as curve.py:244-250 converts them for a Montgomery field). */
#include "oracle_types.h"
#define CURVE nist384
#define PRIME NIST384
#define NL 7
#define NBYTES 48
#define WS_A (-3)
static const spint ws_const_b[7] = {0x8870d0412dcccdu, 0xd9474c32ec0811u, 0x1920022fc429adu, 0x938ae277f2209bu, 0x2094e3374bee94u, 0xf9b62b21f41f02u, 0x8114b604fbfu};
static const spint ws_gen_x[7] = {0x7565fcc0b5284du, 0xe2edd6ce383dd0u, 0x541b4d6e6de378u, 0xa30eff879c3afcu, 0xde2b6454868459u, 0x13812ff723614eu, 0x3aadc2299e15u};
static const spint ws_gen_y[7] = {0x3dad2003a4fe2bu, 0xbfa6b4a9ac2304u, 0x2e83b050ccbfa8u, 0xf4ffd98bade756u, 0xa840c6c3521968u, 0xe9dd8002263969u, 0x78abc25a15c5u};
#include "weierstrass_body.inc"
