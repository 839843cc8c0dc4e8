/* oracle/field_NIST256.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * CPU restatement of what `python3 monty.py 64 NIST256` emits: p = 2^256-2^224+2^192+2^96-1 in
 * Montgomery form, 5 limbs of 52 bits (excess 4), R = 2^260, no virtual limb, ndash = 1 so the
 * reduction digit is simply t & mask (monty.py:2129-2253).  Prime limbs in the generator's signed
 * form (process_prime, monty.py:258-298): [-1, 2^44, 0, 2^36, 0xffffffff0000].
 * Pinned limb-exactly against tests/golden/field_NIST256.json (reference output, see make_golden.py).
 */
#include "oracle_types.h"
#define PRIME NIST256
#define ORACLE_MONTGOMERY
#define NL 5
#define RADIX 52
#define NBITS 256
#define NBYTES 32
#define PM1D2 1
#define P4 ((spint)0xffffffff0000u)
#define PP_CNT 4
static const int pp_idx[PP_CNT] = {0, 1, 3, 4};
static const int pp_sgn[PP_CNT] = {-1, +1, +1, +1};
/* non-trivial root of unity, plain limbs (pseudo.py:1616-1630 / monty.py:2178-2192) */
static const spint roi[NL] = {0xffffffffffffeu, 0xfffffffffffu, 0x0u, 0x1000000000u, 0xffffffff0000u};
static const spint pp_val[PP_CNT] = {1u, (spint)1 << 44, (spint)1 << 36, P4};

void modmul_NIST256(const spint *a, const spint *b, spint *c);
void modsqr_NIST256(const spint *a, spint *c);
void modmli_NIST256(const spint *a, int b, spint *c);
void nres_NIST256(const spint *m, spint *n);
void redc_NIST256(const spint *n, spint *m);
void modpro_NIST256(const spint *w, spint *z);
spint modfsb_NIST256(spint *n);
static spint prop_NIST256(spint *n);

/* Shape-aware reduction terms for column i (mul_process, monty.py:597-627): digit v_j meets prime limb
 * i-j; 2^44 and 2^36 become shifts, p4 a real multiply, the -1 limb is absorbed by v = t & mask.
 * Straight-line columns (oracle/columns.h) so the compiler sees what the generator would emit. */
#include "columns.h"
#define SH44(v) ((dpint)(v) << 44)
#define SH36(v) ((dpint)(v) << 36)
#define MP4(v) ((dpint)(v) * (dpint)P4)
#define NIST256_BODY(COL)                                                                   \
    const spint mask = ((spint)1 << RADIX) - 1;                                             \
    dpint t = 0;                                                                            \
    spint v0, v1, v2, v3, v4;                                                               \
    t += COL(0);                                           v0 = (spint)t & mask; t >>= RADIX; \
    t += COL(1); t += SH44(v0);                            v1 = (spint)t & mask; t >>= RADIX; \
    t += COL(2); t += SH44(v1);                            v2 = (spint)t & mask; t >>= RADIX; \
    t += COL(3); t += SH36(v0); t += SH44(v2);             v3 = (spint)t & mask; t >>= RADIX; \
    t += COL(4); t += MP4(v0); t += SH36(v1); t += SH44(v3); v4 = (spint)t & mask; t >>= RADIX; \
    t += COL(5); t += MP4(v1); t += SH36(v2); t += SH44(v4); c[0] = (spint)t & mask; t >>= RADIX; \
    t += COL(6); t += MP4(v2); t += SH36(v3);              c[1] = (spint)t & mask; t >>= RADIX; \
    t += COL(7); t += MP4(v3); t += SH36(v4);              c[2] = (spint)t & mask; t >>= RADIX; \
    t += COL(8); t += MP4(v4);                             c[3] = (spint)t & mask; t >>= RADIX; \
    c[4] = (spint)t;

/* monty.py:663-872 (non-E branch 840-870), columns getZMU/getZMD 493-537 */
void modmul_NIST256(const spint *a, const spint *b, spint *c) {
#define M(i, j) ((dpint)a[i] * b[j])
#define COL(k) MULCOL5_##k
    NIST256_BODY(COL)
#undef COL
#undef M
}

/* monty.py:982-1165, columns getZSU/getZSD 540-590: tot = 2*sum(cross) + square */
void modsqr_NIST256(const spint *a, spint *c) {
#define S(i, j) ((dpint)a[i] * a[j])
#define COL(k) SQRCOL5_##k
    NIST256_BODY(COL)
#undef COL
#undef S
}

/* Barrett-Dhem branch, monty.py:909-972: r = floor(2^(n+RADIX)/p), h = t >> ((n-64) % RADIX) */
void modmli_NIST256(const spint *a, int b, spint *c) {
    const spint mask = ((spint)1 << RADIX) - 1;
    const spint r = 0x100000000fffffu;
    dpint t = 0;
    for (int i = 0; i < NL - 1; i++) {
        t += (dpint)a[i] * (dpint)b;
        c[i] = (spint)t & mask;
        t >>= RADIX;
    }
    t += (dpint)a[NL - 1] * (dpint)b;
    c[NL - 1] = (spint)t;
    spint h = (spint)(t >> 36);
    spint q = (spint)(((dpint)h * (dpint)r) >> 64);
    c[0] += q;                                                            /* limb -1        */
    t = (dpint)q << 44; c[1] -= (spint)t & mask; c[2] -= (spint)(t >> RADIX); /* limb 2^44  */
    t = (dpint)q << 36; c[3] -= (spint)t & mask; c[4] -= (spint)(t >> RADIX); /* limb 2^36  */
    c[4] -= q * P4;                                                       /* top limb       */
    (void)prop_NIST256(c);
}

/* monty.py:1386-1399: multiply by R^2 mod p */
void nres_NIST256(const spint *m, spint *n) {
    static const spint r2[NL] = {0x300u, 0xffffffff00000u, 0xffffefffffffbu, 0xfdfffffffffffu, 0x4ffffffu};
    modmul_NIST256(m, r2, n);
}

/* monty.py:1402-1416: multiply by 1, final subtract */
void redc_NIST256(const spint *n, spint *m) {
    spint one[NL] = {1, 0, 0, 0, 0};
    modmul_NIST256(n, one, m);
    (void)modfsb_NIST256(m);
}

/* progenitor z = w^PE, PE = (p-3)/4 (monty.py:2158-2165).  Own chain (fixed 4-bit windows over the
 * exponent 0x3fffffffc00000004000000000000000000000003fffffffffffffffffffffff); the reference
 * takes its chain from the external `addchain` tool, so limbs are comparable only after redc. */
void modpro_NIST256(const spint *w, spint *z) {
    static const uint64_t pe[4] = {0xffffffffffffffffULL, 0x000000003fffffffULL, 0x4000000000000000ULL, 0x3fffffffc0000000ULL};
    spint tab[16][NL], acc[NL];
    for (int i = 0; i < NL; i++) tab[1][i] = w[i];
    for (int k = 2; k < 16; k++) modmul_NIST256(tab[k - 1], w, tab[k]);
    int started = 0;
    for (int nib = 63; nib >= 0; nib--) {
        unsigned d = (unsigned)(pe[nib / 16] >> (4 * (nib % 16))) & 15u;
        if (started) for (int s = 0; s < 4; s++) modsqr_NIST256(acc, acc);
        if (d) {
            if (!started) { for (int i = 0; i < NL; i++) acc[i] = tab[d][i]; started = 1; }
            else modmul_NIST256(acc, tab[d], acc);
        }
    }
    for (int i = 0; i < NL; i++) z[i] = acc[i];
}

#include "field_common.inc"
