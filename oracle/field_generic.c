/* oracle/field_generic.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * A run-time interpretation of the reference generators' algorithms for ANY prime they accept in the variants
 * this build supports: instead of emitting specialised C as pseudo.py / monty.py do, the same decisions
 * (pseudo.py:351-728 getZM/getZS/second_pass with the EPM / fred / carry_on / xcess flags; monty.py:493-978
 * mul_process with the 0 / +-1 / 2^k / other limb kinds, the gone_neg borrow convention, the virtual limb, full
 * Montgomery digits, trinomial and Barrett-Dhem modmli) are taken while running, from a parameter block filled
 * by the caller out of the constants CAPTURED FROM THE REFERENCE (tests/golden/field_<P>.json "params").
 * It is slower than the per-prime restatements (field_X25519.c ...) and exists to give the further primes
 * (NIST521, NIST384, group orders, ...) a CPU oracle that is pinned against the reference's golden vectors and
 * can then check GPU batches of any size.  Only tests may use it.
 */
#include "oracle_types.h"
#include <string.h>

#include "field_generic.h"

#define N (P->n)
#define MASK ((((spint)1) << P->radix) - 1)

static spint g_prop(const gparams *P, spint *n) {
    sspint carry = (sspint)n[0];
    carry >>= P->radix;
    n[0] &= MASK;
    for (int i = 1; i < N - 1; i++) {
        carry += (sspint)n[i];
        n[i] = (spint)carry & MASK;
        carry >>= P->radix;
    }
    n[N - 1] += (spint)carry;
    return -((n[N - 1] >> 1) >> 62);
}
static void g_addp(const gparams *P, spint *n, spint x, spint sel) {
    for (int k = 0; k < P->pp_cnt; k++) {
        spint w = (P->pp_val[k] * x) & sel;
        if (P->pp_sgn[k] < 0) n[P->pp_idx[k]] -= w; else n[P->pp_idx[k]] += w;
    }
}
static void g_subp(const gparams *P, spint *n, spint x) {
    for (int k = 0; k < P->pp_cnt; k++) {
        spint w = P->pp_val[k] * x;
        if (P->pp_sgn[k] < 0) n[P->pp_idx[k]] += w; else n[P->pp_idx[k]] -= w;
    }
}
spint gen_flatten(const gparams *P, spint *n) {
    spint carry = g_prop(P, n);
    g_addp(P, n, 1, carry);
    (void)g_prop(P, n);
    return carry & 1;
}
spint gen_modfsb(const gparams *P, spint *n) {
    g_subp(P, n, 1);
    return gen_flatten(P, n);
}
void gen_modadd(const gparams *P, const spint *a, const spint *b, spint *n) {
    for (int i = 0; i < N; i++) n[i] = a[i] + b[i];
    g_subp(P, n, 2);
    spint carry = g_prop(P, n);
    g_addp(P, n, 2, carry);
    (void)g_prop(P, n);
}
void gen_modsub(const gparams *P, const spint *a, const spint *b, spint *n) {
    for (int i = 0; i < N; i++) n[i] = a[i] - b[i];
    spint carry = g_prop(P, n);
    g_addp(P, n, 2, carry);
    (void)g_prop(P, n);
}
void gen_modneg(const gparams *P, const spint *b, spint *n) {
    for (int i = 0; i < N; i++) n[i] = (spint)0 - b[i];
    spint carry = g_prop(P, n);
    g_addp(P, n, 2, carry);
    (void)g_prop(P, n);
}

/* ---------------------------------------------------------------- pseudo-Mersenne (pseudo.py:351-728) */
static void pm_second_pass(const gparams *P, dpint t, spint *v, spint *c) {
    const int R = P->radix, XC = P->xcess;
    spint s, carry;
    if (P->fred) {
        spint ut = (spint)t;
        if (XC > 0) { ut = (ut << XC) + (v[N - 1] >> (R - XC)); v[N - 1] &= (((spint)1) << (R - XC)) - 1; }
        if (P->m > 1) ut *= P->m;
        s = v[0] + (ut & MASK);
        c[0] = s & MASK;
        if (P->carry_on) { ut = (s >> R) + (ut >> R); s = v[1] + (ut & MASK); c[1] = s & MASK; }
        carry = (s >> R) + (ut >> R);
    } else {
        dpint ut = t;
        if (XC > 0) { ut = (ut << XC) + (dpint)(v[N - 1] >> (R - XC)); v[N - 1] &= (((spint)1) << (R - XC)) - 1; }
        if (P->m > 1) ut *= (dpint)P->m;
        s = v[0] + ((spint)ut & MASK);
        c[0] = s & MASK;
        if (P->carry_on) { ut = (dpint)(s >> R) + (ut >> R); s = v[1] + ((spint)ut & MASK); c[1] = s & MASK; }
        carry = (s >> R) + (spint)(ut >> R);
    }
    int k = P->carry_on ? 2 : 1;
    c[k] = v[k] + carry;
    for (int i = k + 1; i < N; i++) c[i] = v[i];
}
static void pm_modmul(const gparams *P, const spint *a, const spint *b, spint *c) {
    dpint t = 0;
    spint v[GMAXN], ma[GMAXN];
    dpint hi = 0;                                   /* spint in the ordinary overflow form, dpint in the bad_overflow one */
    if (P->epm) for (int i = 1; i < N; i++) ma[i] = a[i] * P->mm;
    for (int row = 0; row < N; row++) {
        if (P->epm) {
            for (int k = row + 1; k < N; k++) t += (dpint)ma[k] * (dpint)b[N + row - k];
        } else if (row < N - 1) {
            dpint tt = 0;
            for (int k = row + 1; k < N; k++) tt += (dpint)a[k] * (dpint)b[N + row - k];
            if (P->overflow) {                      /* getZM, pseudo.py:407-420: both bad_overflow_mul forms */
                spint lo = (spint)tt & MASK;
                if (row == 0) t += (dpint)lo * (dpint)P->mm;
                else if (P->bad_overflow) t += (hi + (dpint)lo) * (dpint)P->mm;
                else t += (dpint)(spint)(lo + (spint)hi) * (dpint)P->mm;
                hi = P->bad_overflow ? (tt >> P->radix) : (dpint)(spint)(tt >> P->radix);
            } else {
                tt *= (dpint)P->mm;
                t += tt;
            }
        }
        for (int k = 0; k <= row; k++) t += (dpint)a[k] * (dpint)b[row - k];
        if (row == N - 1 && P->overflow) t += hi * (dpint)P->mm;                 /* pseudo.py:435-436 */
        v[row] = (spint)t & MASK;
        t >>= P->radix;
    }
    pm_second_pass(P, t, v, c);
}
static void pm_modsqr(const gparams *P, const spint *a, spint *c) {
    dpint t = 0;
    spint v[GMAXN], ta[GMAXN], ma[GMAXN];
    dpint hi = 0;
    if (P->epm) for (int i = 1; i < N; i++) { ta[i] = a[i] * (spint)2; ma[i] = a[i] * P->mm; }
    for (int row = 0; row < N; row++) {
        int k = row + 1, l = N - 1;
        if (P->epm) {
            for (; k < l; k++, l--) t += (dpint)ma[k] * (dpint)ta[l];
            if (k == l) t += (dpint)ma[k] * (dpint)a[k];
        } else if (row < N - 1) {
            dpint tt = 0;
            int dble = k < l;
            for (; k < l; k++, l--) tt += (dpint)a[k] * (dpint)a[l];
            if (dble) tt *= 2;
            if (k == l) tt += (dpint)a[k] * (dpint)a[k];
            if (P->overflow) {                      /* getZS, pseudo.py:492-493, 536-550: both bad_overflow_sqr forms */
                spint lo = (spint)tt & MASK;
                if (row == 0) t += (dpint)lo * (dpint)P->mm;
                else if (P->bad_overflow) t += (hi + (dpint)lo) * (dpint)P->mm;
                else t += (dpint)(spint)(lo + (spint)hi) * (dpint)P->mm;
                hi = P->bad_overflow ? (tt >> P->radix) : (dpint)(spint)(tt >> P->radix);
            } else {
                tt *= (dpint)P->mm;
                t += tt;
            }
        } else if (P->overflow) {
            t += hi * (dpint)P->mm;                 /* row N-1: pseudo.py:537-538 */
        }
        k = 0; l = row;
        if (P->epm) {
            for (; k < l; k++, l--) t += (dpint)a[k] * (dpint)ta[l];
            if (k == l) t += (dpint)a[k] * (dpint)a[k];
        } else {
            dpint t2 = 0;
            int dble = k < l;
            for (; k < l; k++, l--) t2 += (dpint)a[k] * (dpint)a[l];
            if (dble) t2 *= 2;
            if (k == l) t2 += (dpint)a[k] * (dpint)a[k];
            t += t2;
        }
        v[row] = (spint)t & MASK;
        t >>= P->radix;
    }
    pm_second_pass(P, t, v, c);
}
static void pm_modmli(const gparams *P, const spint *a, int b, spint *c) {
    dpint t = 0;
    spint v[GMAXN];
    for (int i = 0; i < N; i++) {
        t += (dpint)a[i] * (dpint)b;
        v[i] = (spint)t & MASK;
        t >>= P->radix;
    }
    pm_second_pass(P, t, v, c);
}

/* ---------------------------------------------------------------- Montgomery (monty.py:493-978) */
static int is_pow2(long long d) { return d > 1 && (d & (d - 1)) == 0; }
static int log2ll(long long d) { int e = 0; while (((long long)1 << e) < d) e++; return e; }

/* reduction contribution of column c: digit v_j meets signed prime limb l = c - j (mul_process, 597-627) */
/* monty.py's PM form (an exploitable pseudo-Mersenne 2^n - M given to monty.py: ppw[0] = -M, monty.py:284-288): the low limb is
 * carried with the borrow technique scaled by M -- column 0 adds M*(q - v0), every later column adds M*mask and takes M*v_i
 * back with its own digit, and the last limb gives back M (monty.py:700-870) */
static spint pm_m(const gparams *P) { return (P->family && P->ppw[0] < -1) ? (spint)(-P->ppw[0]) : 0; }
static void mo_reduce(const gparams *P, int c, dpint *t, const spint *v) {
    const spint q = ((spint)1) << P->radix, mask = q - 1;
    if (pm_m(P) && c >= 1) *t += (dpint)(spint)(pm_m(P) * mask);
    const int lmax = P->E ? N : N - 1;
    const int scratch = P->neg_limb > 0 && c > P->neg_limb;
    spint s = mask;
    for (int l = 1; l <= lmax; l++) {
        int j = c - l;
        if (j < 0 || j > lmax || j >= c) continue;
        long long d = P->ppw[l];
        if (d > 1) {
            if (is_pow2(d)) *t += (dpint)v[j] << log2ll(d);
            else *t += (dpint)v[j] * (dpint)(spint)d;
        } else if (d == 1) {
            if (scratch) s += v[j]; else *t += (dpint)v[j];
        } else if (d == -1) {
            if (scratch) s -= v[j]; else *t += (dpint)(spint)(q - v[j]);
        }
    }
    if (scratch) *t += (dpint)s;
}
static spint mo_digit(const gparams *P, dpint *t, int col) {
    if (P->ndash == 1) return (spint)*t & MASK;
    spint v = ((spint)*t * P->ndash) & MASK;
    if (pm_m(P)) {
        const spint q = ((spint)1) << P->radix;
        if (col == 0) *t += (dpint)(spint)(pm_m(P) * (q - v)); else *t -= (dpint)(spint)(pm_m(P) * v);
    } else if (P->ppw[0] == 1) *t += (dpint)v; else *t += (dpint)v * (dpint)(spint)P->ppw[0];
    return v;
}
static void mo_mul(const gparams *P, const spint *a, const spint *b, spint *c, int sqr) {
    const int jmax = P->E ? N : N - 1;
    const int ncol = P->E ? 2 * N : 2 * N - 1;
    dpint t = 0;
    spint v[GMAXN + 1];
    for (int col = 0; col < ncol; col++) {
        int lo = col < N ? 0 : col - (N - 1), hi = col < N ? col : N - 1;
        if (lo <= hi) {
            if (!sqr) {
                for (int k = lo; k <= hi; k++) t += (dpint)a[k] * b[col - k];
            } else {
                dpint tot = 0;
                int k = lo, hap = 0;
                for (; k < col - k; k++) { tot += (dpint)a[k] * a[col - k]; hap = 1; }
                if (hap) tot *= 2;
                if (col % 2 == 0) tot += (dpint)a[col / 2] * a[col / 2];
                t += tot;
            }
        }
        mo_reduce(P, col, &t, v);
        if (col <= jmax) v[col] = mo_digit(P, &t, col);
        else c[col - jmax - 1] = (spint)t & MASK;
        t >>= P->radix;
    }
    if (P->E) {
        if (pm_m(P)) t += (dpint)(spint)(v[N] - pm_m(P));
        else if (P->neg_limb > 0) t += (dpint)(spint)(v[N] - (spint)1); else t += (dpint)v[N];
    } else if (pm_m(P)) {
        t -= (dpint)pm_m(P);
    } else if (P->neg_limb > 0) {
        t -= (dpint)1;
    }
    c[N - 1] = (spint)t;
}
static void mo_modmli(const gparams *P, const spint *a, int b, spint *c) {
    const int R = P->radix;
    dpint t = 0;
    if (P->trin > 0) {
        for (int i = 0; i < N; i++) { t += (dpint)a[i] * (dpint)b; c[i] = (spint)t & MASK; t >>= R; }
        spint s = (spint)t;
        if (P->xcess > 0) { s = (s << P->xcess) + (c[N - 1] >> (R - P->xcess)); c[N - 1] &= (((spint)1) << (R - P->xcess)) - 1; }
        c[0] += s;
        c[P->trin] += s;
        return;
    }
    for (int i = 0; i < N - 1; i++) { t += (dpint)a[i] * (dpint)b; c[i] = (spint)t & MASK; t >>= R; }
    t += (dpint)a[N - 1] * (dpint)b;
    c[N - 1] = (spint)t;
    spint h = (spint)(t >> ((P->nbits - 64) % R));
    spint q = (spint)(((dpint)h * (dpint)P->barrett_r) >> 64);
    int propc = P->ppw[0] > 0;
    for (int i = 0; i < N; i++) {
        long long d = P->ppw[i];
        if (i >= 1 && i < N - 1 && d != 0) propc = 1;
        if (d == 0) continue;
        if (d == -1) c[i] += q;
        else if (d == 1) c[i] -= q;
        else if (d < 0) { dpint w = (dpint)q * (dpint)(spint)(-d); c[i] += (spint)w & MASK; c[i + 1] += (spint)(w >> R); }   /* PM: limb 0 */
        else if (is_pow2(d)) {
            if (i < N - 1) { dpint w = (dpint)q << log2ll(d); c[i] -= (spint)w & MASK; c[i + 1] -= (spint)(w >> R); }
            else c[i] -= q << log2ll(d);
        } else {
            if (i < N - 1) { dpint w = (dpint)q * (dpint)(spint)d; c[i] -= (spint)w & MASK; c[i + 1] -= (spint)(w >> R); }
            else c[i] -= q * (spint)d;
        }
    }
    if (P->E) c[N - 1] -= q << R;
    if (propc) (void)g_prop(P, c);
}

/* ---------------------------------------------------------------- family dispatch and the rest of the API */
void gen_modmul(const gparams *P, const spint *a, const spint *b, spint *c) {
    spint x[GMAXN], y[GMAXN];
    memcpy(x, a, sizeof(spint) * N); memcpy(y, b, sizeof(spint) * N);      /* aliasing-safe */
    if (P->family) mo_mul(P, x, y, c, 0); else pm_modmul(P, x, y, c);
}
void gen_modsqr(const gparams *P, const spint *a, spint *c) {
    spint x[GMAXN];
    memcpy(x, a, sizeof(spint) * N);
    if (P->family) mo_mul(P, x, x, c, 1); else pm_modsqr(P, x, c);
}
void gen_modmli(const gparams *P, const spint *a, int b, spint *c) {
    spint x[GMAXN];
    memcpy(x, a, sizeof(spint) * N);
    if (P->family) mo_modmli(P, x, b, c); else pm_modmli(P, x, b, c);
}
void gen_nres(const gparams *P, const spint *m, spint *n) {
    if (P->family) gen_modmul(P, m, P->r2, n); else memmove(n, m, sizeof(spint) * N);
}
void gen_redc(const gparams *P, const spint *n, spint *m) {
    if (P->family) {
        spint one[GMAXN] = {1};
        gen_modmul(P, n, one, m);
    } else {
        memmove(m, n, sizeof(spint) * N);
    }
    (void)gen_modfsb(P, m);
}
/* progenitor x^PE by square-and-multiply over the exponent bits (the reference uses an addchain chain) */
void gen_modpro(const gparams *P, const spint *w, spint *z) {
    spint x[GMAXN], acc[GMAXN];
    memcpy(x, w, sizeof(spint) * N);
    int top = 64 * P->pe_words - 1;
    while (top > 0 && !((P->pe[top / 64] >> (top % 64)) & 1)) top--;
    memcpy(acc, x, sizeof(spint) * N);
    for (int i = top - 1; i >= 0; i--) {
        gen_modsqr(P, acc, acc);
        if ((P->pe[i / 64] >> (i % 64)) & 1) gen_modmul(P, acc, x, acc);
    }
    memcpy(z, acc, sizeof(spint) * N);
}
void gen_modinv(const gparams *P, const spint *x, const spint *h, spint *z) {
    spint s[GMAXN], t[GMAXN], xx[GMAXN];
    memcpy(xx, x, sizeof(spint) * N);
    if (h == NULL) gen_modpro(P, xx, t); else memcpy(t, h, sizeof(spint) * N);
    memcpy(s, xx, sizeof(spint) * N);
    for (int i = 0; i < P->pm1d2 - 1; i++) { gen_modsqr(P, s, s); gen_modmul(P, s, xx, s); }
    for (int i = 0; i < P->pm1d2 + 1; i++) gen_modsqr(P, t, t);
    gen_modmul(P, s, t, z);
}
int gen_modis1(const gparams *P, const spint *a) {
    spint c[GMAXN], d = 0;
    gen_redc(P, a, c);
    for (int i = 1; i < N; i++) d |= c[i];
    return (int)((spint)1 & ((d - (spint)1) >> P->radix) & (((c[0] ^ (spint)1) - (spint)1) >> P->radix));
}
int gen_modis0(const gparams *P, const spint *a) {
    spint c[GMAXN], d = 0;
    gen_redc(P, a, c);
    for (int i = 0; i < N; i++) d |= c[i];
    return (int)((spint)1 & ((d - (spint)1) >> P->radix));
}
int gen_modqr(const gparams *P, const spint *h, const spint *x) {
    spint r[GMAXN];
    if (h == NULL) { gen_modpro(P, x, r); gen_modsqr(P, r, r); } else gen_modsqr(P, h, r);
    gen_modmul(P, r, x, r);
    for (int i = 0; i < P->pm1d2 - 1; i++) gen_modsqr(P, r, r);
    return gen_modis1(P, r) | gen_modis0(P, x);
}
void gen_modsqrt(const gparams *P, const spint *x, const spint *h, spint *r) {
    spint s[GMAXN], y[GMAXN], xx[GMAXN];
    memcpy(xx, x, sizeof(spint) * N);
    if (h == NULL) gen_modpro(P, xx, y); else memcpy(y, h, sizeof(spint) * N);
    gen_modmul(P, y, xx, s);
    if (P->pm1d2 > 1) {
        spint t[GMAXN], b[GMAXN], v[GMAXN], z[GMAXN];
        memcpy(z, P->roi, sizeof(spint) * N);
        gen_modmul(P, s, y, t);
        gen_nres(P, z, z);
        for (int k = P->pm1d2; k > 1; k--) {
            memcpy(b, t, sizeof(spint) * N);
            for (int i = 0; i < k - 2; i++) gen_modsqr(P, b, b);
            int d = 1 - gen_modis1(P, b);
            gen_modmul(P, s, z, v);
            if (d) memcpy(s, v, sizeof(spint) * N);
            gen_modsqr(P, z, z);
            gen_modmul(P, t, z, v);
            if (d) memcpy(t, v, sizeof(spint) * N);
        }
    }
    memcpy(r, s, sizeof(spint) * N);
}

/* batched views over SoA buffers buf[limb*ld + j]; op: 0 modmul 1 modadd 2 modsub 3 modsqr 4 modneg 5 nres 6 redc
 * 7 modinv 8 modsqrt */
void gen_batch(const gparams *P, int op, const spint *a, const spint *b, spint *c, size_t n, size_t ld) {
    for (size_t j = 0; j < n; j++) {
        spint x[GMAXN], y[GMAXN], z[GMAXN];
        for (int i = 0; i < N; i++) { x[i] = a[(size_t)i * ld + j]; y[i] = b ? b[(size_t)i * ld + j] : 0; }
        switch (op) {
        case 0: gen_modmul(P, x, y, z); break;
        case 1: gen_modadd(P, x, y, z); break;
        case 2: gen_modsub(P, x, y, z); break;
        case 3: gen_modsqr(P, x, z); break;
        case 4: gen_modneg(P, x, z); break;
        case 5: gen_nres(P, x, z); break;
        case 6: gen_redc(P, x, z); break;
        case 7: gen_modinv(P, x, NULL, z); break;
        case 8: gen_modsqrt(P, x, NULL, z); break;
        default: memset(z, 0, sizeof z);
        }
        for (int i = 0; i < N; i++) c[(size_t)i * ld + j] = z[i];
    }
}
void gen_batch_mli(const gparams *P, const spint *a, int k, spint *c, size_t n, size_t ld) {
    for (size_t j = 0; j < n; j++) {
        spint x[GMAXN], z[GMAXN];
        for (int i = 0; i < N; i++) x[i] = a[(size_t)i * ld + j];
        gen_modmli(P, x, k, z);
        for (int i = 0; i < N; i++) c[(size_t)i * ld + j] = z[i];
    }
}
size_t gen_params_size(void) { return sizeof(gparams); }
