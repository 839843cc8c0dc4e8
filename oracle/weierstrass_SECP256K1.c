/* oracle/weierstrass_SECP256K1.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 * secp256k1 instance of the Weierstrass-layer restatement: a = 0, b = 7 kept as the C int CONSTANT_B (curve.py:190-198,
 * 235-238), generator as plain limbs (pseudo-Mersenne field). */
#include "oracle_types.h"
#define CURVE secp256k1
#define PRIME SECP256K1
#define NL 5
#define NBYTES 32
#define WS_A 0
#define WS_SMALL_B 7
static const spint ws_gen_x[5] = {0x2815b16f81798u, 0xdb2dce28d959fu, 0xe870b07029bfcu, 0xbbac55a06295cu, 0x79be667ef9dcu};
static const spint ws_gen_y[5] = {0x7d08ffb10d4b8u, 0x48a68554199c4u, 0xe1108a8fd17b4u, 0xc4655da4fbfc0u, 0x483ada7726a3u};
#include "weierstrass_body.inc"
