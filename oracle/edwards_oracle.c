/* oracle/edwards_oracle.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 * ED25519 and ED448 instances of the Edwards-layer restatement (edwards_body.inc).  Curve constants are the
 * ones curve.py:85-105 lists, converted to internal-form limbs as curve.py:244-298 does (plain limbs for the
 * pseudo-Mersenne field, value*R mod p for the Montgomery field, small B kept as an int). */
#include "oracle_types.h"

/* ---- ED25519: -x^2 + y^2 = 1 + d x^2 y^2 over 2^255-19 (pseudo.py field, 5 x 51) */
#define CURVE ed25519
#define PRIME X25519
#define NL 5
#define NBYTES 32
#define ED_A (-1)
#define ED_COF 3
static const spint ed_const_b[5] = {0x34dca135978a3u, 0x1a8283b156ebdu, 0x5e7a26001c029u, 0x739c663a03cbbu, 0x52036cee2b6ffu};
static const spint ed_gen_x[5] = {0x62d608f25d51au, 0x412a4b4f6592au, 0x75b7171a4b31du, 0x1ff60527118feu, 0x216936d3cd6e5u};
static const spint ed_gen_y[5] = {0x6666666666658u, 0x4ccccccccccccu, 0x1999999999999u, 0x3333333333333u, 0x6666666666666u};
#include "edwards_body.inc"
#undef CURVE
#undef PRIME
#undef NL
#undef NBYTES
#undef ED_A
#undef ED_COF
#define ed_const_b ed448_unused_b
#define ed_gen_x ed448_gen_x
#define ed_gen_y ed448_gen_y

/* ---- ED448: x^2 + y^2 = 1 - 39081 x^2 y^2 over 2^448-2^224-1 (monty.py field, 8 x 56, R = 2^504) */
#define CURVE ed448
#define PRIME X448
#define NL 8
#define NBYTES 56
#define ED_A 1
#define ED_COF 2
#define ED_B_SMALL (-39081)
static const spint ed448_gen_x[8] = {0x420685f0ea8836u, 0x35bf93b17aa383u, 0xb7bc2914f8fe6du, 0xe44cd37ab765fau, 0x34f39b1b69235eu, 0x44d6fb9be886a8u, 0xee96c7295e6eb4u, 0xd16ef0905d88b9u};
static const spint ed448_gen_y[8] = {0xd81f4fba184177u, 0xac119c79a99632u, 0xda8e9ac23c2104u, 0x416ef259fc5486u, 0x46ff5902c1cc32u, 0x4fa9dd01223251u, 0xa1f0e6acaf9471u, 0x65f7687a33ab50u};
#include "edwards_body.inc"
