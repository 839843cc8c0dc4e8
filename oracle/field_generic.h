/* oracle/field_generic.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 * Parameter block and entry points of the run-time generic oracle (field_generic.c). */
#ifndef ORACLE_FIELD_GENERIC_H
#define ORACLE_FIELD_GENERIC_H
#include "oracle_types.h"
#include <stddef.h>

#define GMAXN 16

typedef struct {
    int family;          /* 0 = pseudo-Mersenne, 1 = Montgomery */
    int n, radix, nbits, nbytes, xcess, pm1d2;
    /* pseudo-Mersenne */
    spint m, mm;
    int epm, fred, carry_on;
    /* Montgomery */
    long long ppw[GMAXN + 1];   /* signed prime limbs, + virtual limb if E */
    int E, trin, neg_limb;
    spint ndash, barrett_r;
    spint r2[GMAXN];
    /* caddp / addp / subp */
    int pp_cnt, pp_idx[GMAXN], pp_sgn[GMAXN];
    spint pp_val[GMAXN];
    /* progenitor exponent, little-endian 64-bit words, and a 2^pm1d2-th root of unity (plain limbs) */
    int pe_words;
    spint pe[GMAXN];
    spint roi[GMAXN];
    /* pseudo.py:1640-1657: column sums would overflow 128 bits, the folded high part is split (lo, hi) instead */
    int overflow;
    int bad_overflow;    /* pseudo.py:1646-1648: the carried high part needs a double word too (bad_overflow_mul / _sqr) */
} gparams;

spint gen_flatten(const gparams *P, spint *n);
spint gen_modfsb(const gparams *P, spint *n);
void gen_modadd(const gparams *P, const spint *a, const spint *b, spint *n);
void gen_modsub(const gparams *P, const spint *a, const spint *b, spint *n);
void gen_modneg(const gparams *P, const spint *b, spint *n);
void gen_modmul(const gparams *P, const spint *a, const spint *b, spint *c);
void gen_modsqr(const gparams *P, const spint *a, spint *c);
void gen_modmli(const gparams *P, const spint *a, int b, spint *c);
void gen_nres(const gparams *P, const spint *m, spint *n);
void gen_redc(const gparams *P, const spint *n, spint *m);
void gen_modpro(const gparams *P, const spint *w, spint *z);
void gen_modinv(const gparams *P, const spint *x, const spint *h, spint *z);
int gen_modis1(const gparams *P, const spint *a);
int gen_modis0(const gparams *P, const spint *a);
int gen_modqr(const gparams *P, const spint *h, const spint *x);
void gen_modsqrt(const gparams *P, const spint *x, const spint *h, spint *r);
#endif
