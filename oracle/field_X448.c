/* oracle/field_X448.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * CPU restatement of what `python3 monty.py 64 X448` emits: p = 2^448-2^224-1 in Montgomery form,
 * 8 limbs of 56 bits, excess 0 so a virtual 9th limb is added and R = 2^504; signed prime limbs
 * [-1,0,0,0,-1,0,0,0,+1], ndash = 1, "lucky trinomial" with trin = 4 (monty.py:2129-2253).
 * Pinned limb-exactly against tests/golden/field_X448.json (reference output, see make_golden.py).
 */
#include "oracle_types.h"
#define PRIME X448
#define ORACLE_MONTGOMERY
#define NL 8
#define RADIX 56
#define NBITS 448
#define NBYTES 56
#define PM1D2 1
#define PP_CNT 3
/* virtual limb (+1 at index 8) is applied as +2^56 at limb 7 (caddp "if E", monty.py:313-314) */
static const int pp_idx[PP_CNT] = {0, 4, 7};
static const int pp_sgn[PP_CNT] = {-1, -1, +1};
/* non-trivial root of unity, plain limbs (pseudo.py:1616-1630 / monty.py:2178-2192) */
static const spint roi[NL] = {0xfffffffffffffeu, 0xffffffffffffffu, 0xffffffffffffffu, 0xffffffffffffffu, 0xfffffffffffffeu, 0xffffffffffffffu, 0xffffffffffffffu, 0xffffffffffffffu};
static const spint pp_val[PP_CNT] = {1u, 1u, (spint)1 << 56};

void modmul_X448(const spint *a, const spint *b, spint *c);
void modsqr_X448(const spint *a, spint *c);
void modmli_X448(const spint *a, int b, spint *c);
void nres_X448(const spint *m, spint *n);
void redc_X448(const spint *n, spint *m);
void modpro_X448(const spint *w, spint *z);
spint modfsb_X448(spint *n);

/* Reduction terms (mul_process with the gone_neg/mask_set borrow convention, monty.py:597-627, 717-738,
 * 778-838): nine digits v0..v8 meet prime limbs -1 (index 0, implicit), -1 (index 4) and +1 (index 8).
 * The first negative use adds q - v0; every later column starts a 64-bit scratch at mask (= q-1, carrying
 * the outstanding borrow), adds v_{i-8} and subtracts v_{i-4}.  Straight-line columns (oracle/columns.h). */
#include "columns.h"
#define X448_BODY(COL)                                                                             \
    const spint q = (spint)1 << RADIX, mask = q - 1;                                               \
    dpint t = 0;                                                                                   \
    spint v0, v1, v2, v3, v4, v5, v6, v7, v8, s;                                                   \
    t += COL(0);                                        v0 = (spint)t & mask; t >>= RADIX;         \
    t += COL(1);                                        v1 = (spint)t & mask; t >>= RADIX;         \
    t += COL(2);                                        v2 = (spint)t & mask; t >>= RADIX;         \
    t += COL(3);                                        v3 = (spint)t & mask; t >>= RADIX;         \
    t += COL(4);  t += (dpint)(spint)(q - v0);          v4 = (spint)t & mask; t >>= RADIX;         \
    t += COL(5);  s = mask; s -= v1;           t += (dpint)s; v5 = (spint)t & mask; t >>= RADIX;   \
    t += COL(6);  s = mask; s -= v2;           t += (dpint)s; v6 = (spint)t & mask; t >>= RADIX;   \
    t += COL(7);  s = mask; s -= v3;           t += (dpint)s; v7 = (spint)t & mask; t >>= RADIX;   \
    t += COL(8);  s = mask; s += v0; s -= v4;  t += (dpint)s; v8 = (spint)t & mask; t >>= RADIX;   \
    t += COL(9);  s = mask; s += v1; s -= v5;  t += (dpint)s; c[0] = (spint)t & mask; t >>= RADIX; \
    t += COL(10); s = mask; s += v2; s -= v6;  t += (dpint)s; c[1] = (spint)t & mask; t >>= RADIX; \
    t += COL(11); s = mask; s += v3; s -= v7;  t += (dpint)s; c[2] = (spint)t & mask; t >>= RADIX; \
    t += COL(12); s = mask; s += v4; s -= v8;  t += (dpint)s; c[3] = (spint)t & mask; t >>= RADIX; \
    t += COL(13); s = mask; s += v5;           t += (dpint)s; c[4] = (spint)t & mask; t >>= RADIX; \
    t += COL(14); s = mask; s += v6;           t += (dpint)s; c[5] = (spint)t & mask; t >>= RADIX; \
    s = mask; s += v7;                         t += (dpint)s; c[6] = (spint)t & mask; t >>= RADIX; \
    t += (dpint)(spint)(v8 - (spint)1);        /* settle the borrow, monty.py:830-838 */           \
    c[7] = (spint)t;

/* monty.py:663-872, E branch 778-838 */
void modmul_X448(const spint *a, const spint *b, spint *c) {
#define M(i, j) ((dpint)a[i] * b[j])
#define COL(k) MULCOL8_##k
    X448_BODY(COL)
#undef COL
#undef M
}

/* monty.py:982-1165 */
void modsqr_X448(const spint *a, spint *c) {
#define S(i, j) ((dpint)a[i] * a[j])
#define COL(k) SQRCOL8_##k
    X448_BODY(COL)
#undef COL
#undef S
}

/* trinomial branch, monty.py:888-907: fold the overflow word into limbs 0 and trin */
void modmli_X448(const spint *a, int b, spint *c) {
    const spint mask = ((spint)1 << RADIX) - 1;
    dpint t = 0;
    for (int i = 0; i < NL; i++) {
        t += (dpint)a[i] * (dpint)b;
        c[i] = (spint)t & mask;
        t >>= RADIX;
    }
    spint s = (spint)t;
    c[0] += s;
    c[4] += s;
}

/* monty.py:1386-1399: R^2 mod p = 2 * 2^112 + 3 * 2^336 */
void nres_X448(const spint *m, spint *n) {
    static const spint r2[NL] = {0, 0, 2, 0, 0, 0, 3, 0};
    modmul_X448(m, r2, n);
}

/* monty.py:1402-1416 */
void redc_X448(const spint *n, spint *m) {
    spint one[NL] = {1, 0, 0, 0, 0, 0, 0, 0};
    modmul_X448(n, one, m);
    (void)modfsb_X448(m);
}

/* progenitor z = w^PE, PE = (p-3)/4 = 2^446 - 2^222 - 1 = (2^223-1)*2^223 + (2^222-1).
 * Own chain: run ladder to 2^222-1 (1,2,3,6,12,24,27,54,108,111,222), one more step to 2^223-1. */
static void sqn448(const spint *a, int n, spint *c) {
    modsqr_X448(a, c);
    for (int i = 1; i < n; i++) modsqr_X448(c, c);
}
void modpro_X448(const spint *w, spint *z) {
    spint x[NL], a[NL], b[NL], d[NL], e[NL];
    for (int i = 0; i < NL; i++) x[i] = w[i];
    sqn448(x, 1, a);   modmul_X448(a, x, a);   /* 2^2-1   */
    sqn448(a, 1, b);   modmul_X448(b, x, b);   /* 2^3-1   */
    sqn448(b, 3, a);   modmul_X448(a, b, a);   /* 2^6-1   */
    sqn448(a, 6, d);   modmul_X448(d, a, d);   /* 2^12-1  */
    sqn448(d, 12, a);  modmul_X448(a, d, a);   /* 2^24-1  */
    sqn448(a, 3, a);   modmul_X448(a, b, a);   /* 2^27-1  */
    sqn448(a, 27, d);  modmul_X448(d, a, d);   /* 2^54-1  */
    sqn448(d, 54, a);  modmul_X448(a, d, a);   /* 2^108-1 */
    sqn448(a, 3, a);   modmul_X448(a, b, a);   /* 2^111-1 */
    sqn448(a, 111, d); modmul_X448(d, a, d);   /* 2^222-1 */
    sqn448(d, 1, e);   modmul_X448(e, x, e);   /* 2^223-1 */
    sqn448(e, 223, e); modmul_X448(e, d, z);   /* PE      */
}

#include "field_common.inc"
