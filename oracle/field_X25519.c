/* oracle/field_X25519.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * CPU restatement of what `python3 pseudo.py 64 X25519` emits (p = 2^255-19, 5 limbs of 51 bits,
 * excess 0, mm = 19, flags EPM=True fred=True overflow=False carry_on=False; pseudo.py:1561-1678).
 * Pinned limb-exactly against tests/golden/field_X25519.json, which was produced by running the
 * reference generator in the build container (tests/golden/make_golden.py).
 * modpro uses our own addition chain (the reference shells out to `addchain`, pseudo.py:1582-1587),
 * so modpro/modinv limbs are only comparable after redc.
 */
#include "oracle_types.h"
#define PRIME X25519
#define NL 5
#define RADIX 51
#define NBITS 255
#define NBYTES 32
#define PM1D2 2        /* p-1 = 2^2 * odd (pseudo.py:1575-1581) */
#define MM 19          /* m * 2^xcess (pseudo.py:1596) */
#define PP_CNT 2
/* p = -19 + 2^51 * 2^(51*4): caddp/addp/subp touch limb 0 and limb N-1 (pseudo.py:202-220) */
static const int pp_idx[PP_CNT] = {0, 4};
static const int pp_sgn[PP_CNT] = {-1, +1};
/* non-trivial root of unity, plain limbs (pseudo.py:1616-1630 / monty.py:2178-2192) */
static const spint roi[NL] = {0x61b274a0ea0b0u, 0xd5a5fc8f189du, 0x7ef5e9cbd0c60u, 0x78595a6804c9eu, 0x2b8324804fc1du};
static const spint pp_val[PP_CNT] = {19u, (spint)1 << 51};

void modmul_X25519(const spint *a, const spint *b, spint *c);
void modsqr_X25519(const spint *a, spint *c);
void modmli_X25519(const spint *a, int b, spint *c);
void nres_X25519(const spint *m, spint *n);
void redc_X25519(const spint *n, spint *m);
void modpro_X25519(const spint *w, spint *z);
spint modfsb_X25519(spint *n);

/* second reduction pass, fred branch with xcess=0, m>1, no carry_on (pseudo.py:557-611) */
static inline void second_pass(dpint t, const spint *v, spint *c) {
    const spint mask = ((spint)1 << RADIX) - 1;
    spint ut = (spint)t;
    ut *= MM;
    spint s = v[0] + (ut & mask);
    c[0] = s & mask;
    spint carry = (s >> RADIX) + (ut >> RADIX);
    c[1] = v[1] + carry;
    for (int i = 2; i < NL; i++) c[i] = v[i];
}

/* Comba rows with the high half pre-multiplied by mm (EPM): pseudo.py:616-659, getZM 390-438.
 * Row r: t += sum_{k>r} ma_k*b_{5+r-k} + sum_{k<=r} a_k*b_{r-k}.  Straight-line, as the generator emits. */
void modmul_X25519(const spint *a, const spint *b, spint *c) {
    const spint mask = ((spint)1 << RADIX) - 1;
    const spint ma1 = a[1] * (spint)MM, ma2 = a[2] * (spint)MM, ma3 = a[3] * (spint)MM, ma4 = a[4] * (spint)MM;
    dpint t = 0;
    spint v[NL];
#define W(x, y) ((dpint)(x) * (dpint)(y))
    t += W(ma1, b[4]) + W(ma2, b[3]) + W(ma3, b[2]) + W(ma4, b[1]) + W(a[0], b[0]);
    v[0] = (spint)t & mask; t >>= RADIX;
    t += W(ma2, b[4]) + W(ma3, b[3]) + W(ma4, b[2]) + W(a[0], b[1]) + W(a[1], b[0]);
    v[1] = (spint)t & mask; t >>= RADIX;
    t += W(ma3, b[4]) + W(ma4, b[3]) + W(a[0], b[2]) + W(a[1], b[1]) + W(a[2], b[0]);
    v[2] = (spint)t & mask; t >>= RADIX;
    t += W(ma4, b[4]) + W(a[0], b[3]) + W(a[1], b[2]) + W(a[2], b[1]) + W(a[3], b[0]);
    v[3] = (spint)t & mask; t >>= RADIX;
    t += W(a[0], b[4]) + W(a[1], b[3]) + W(a[2], b[2]) + W(a[3], b[1]) + W(a[4], b[0]);
    v[4] = (spint)t & mask; t >>= RADIX;
    second_pass(t, v, c);
}

/* squaring rows with ta=2a, ma=19a (EPM): pseudo.py:663-702, getZS 441-554 */
void modsqr_X25519(const spint *a, spint *c) {
    const spint mask = ((spint)1 << RADIX) - 1;
    const spint ta1 = a[1] * (spint)2, ta2 = a[2] * (spint)2, ta3 = a[3] * (spint)2, ta4 = a[4] * (spint)2;
    const spint ma1 = a[1] * (spint)MM, ma2 = a[2] * (spint)MM, ma3 = a[3] * (spint)MM, ma4 = a[4] * (spint)MM;
    dpint t = 0;
    spint v[NL];
    t += W(ma1, ta4) + W(ma2, ta3) + W(a[0], a[0]);
    v[0] = (spint)t & mask; t >>= RADIX;
    t += W(ma2, ta4) + W(ma3, a[3]) + W(a[0], ta1);
    v[1] = (spint)t & mask; t >>= RADIX;
    t += W(ma3, ta4) + W(a[0], ta2) + W(a[1], a[1]);
    v[2] = (spint)t & mask; t >>= RADIX;
    t += W(ma4, a[4]) + W(a[0], ta3) + W(a[1], ta2);
    v[3] = (spint)t & mask; t >>= RADIX;
    t += W(a[0], ta4) + W(a[1], ta3) + W(a[2], a[2]);
    v[4] = (spint)t & mask; t >>= RADIX;
    second_pass(t, v, c);
#undef W
}

/* pseudo.py:705-728; (dpint)b of a negative int sign-extends exactly as in the emitted C */
void modmli_X25519(const spint *a, int b, spint *c) {
    const spint mask = ((spint)1 << RADIX) - 1;
    dpint t = 0;
    spint v[NL];
    for (int i = 0; i < NL; i++) {
        t += (dpint)a[i] * (dpint)b;
        v[i] = (spint)t & mask;
        t >>= RADIX;
    }
    second_pass(t, v, c);
}

/* pseudo.py:952-962: identity */
void nres_X25519(const spint *m, spint *n) {
    for (int i = 0; i < NL; i++) n[i] = m[i];
}

/* pseudo.py:965-976: copy + final subtract */
void redc_X25519(const spint *n, spint *m) {
    for (int i = 0; i < NL; i++) m[i] = n[i];
    (void)modfsb_X25519(m);
}

/* progenitor z = w^PE, PE = (p-5)/8 = 2^252-3 (pseudo.py:1575-1581, 758-785).
 * Own chain: 2^250-1 by the usual 1,2,4,5,10,20,40,50,100,200,250 run ladder, then *4+1. */
static void sqn(const spint *a, int n, spint *c) {
    modsqr_X25519(a, c);
    for (int i = 1; i < n; i++) modsqr_X25519(c, c);
}
void modpro_X25519(const spint *w, spint *z) {
    spint x[NL], t0[NL], t1[NL], t2[NL], t3[NL];
    for (int i = 0; i < NL; i++) x[i] = w[i];
    sqn(x, 1, t0);            modmul_X25519(t0, x, t0);   /* 2^2-1   */
    sqn(t0, 2, t1);           modmul_X25519(t1, t0, t1);  /* 2^4-1   */
    sqn(t1, 1, t1);           modmul_X25519(t1, x, t1);   /* 2^5-1   */
    sqn(t1, 5, t2);           modmul_X25519(t2, t1, t2);  /* 2^10-1  */
    sqn(t2, 10, t3);          modmul_X25519(t3, t2, t3);  /* 2^20-1  */
    sqn(t3, 20, t0);          modmul_X25519(t0, t3, t0);  /* 2^40-1  */
    sqn(t0, 10, t0);          modmul_X25519(t0, t2, t0);  /* 2^50-1  */
    sqn(t0, 50, t2);          modmul_X25519(t2, t0, t2);  /* 2^100-1 */
    sqn(t2, 100, t3);         modmul_X25519(t3, t2, t3);  /* 2^200-1 */
    sqn(t3, 50, t3);          modmul_X25519(t3, t0, t3);  /* 2^250-1 */
    sqn(t3, 2, t3);           modmul_X25519(t3, x, z);    /* 2^252-3 */
}

#include "field_common.inc"
