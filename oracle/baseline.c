/* oracle/baseline.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 * Multi-threaded drivers used only by bench.py's cpu_baseline leg: the scalar oracle functions
 * applied to SoA batches / AoS ladder records, one contiguous slice per pthread. */
#include "oracle_types.h"
#include <pthread.h>

void batch_modmul_X25519(const spint *, const spint *, spint *, size_t, size_t);
void batch_modmul_NIST256(const spint *, const spint *, spint *, size_t, size_t);
void batch_modmul_X448(const spint *, const spint *, spint *, size_t, size_t);
void batch_modsqr_X25519(const spint *, spint *, size_t, size_t);
void batch_modsqr_NIST256(const spint *, spint *, size_t, size_t);
void batch_modsqr_X448(const spint *, spint *, size_t, size_t);
void batch_nres_X25519(const spint *, spint *, size_t, size_t);
void batch_nres_NIST256(const spint *, spint *, size_t, size_t);
void batch_nres_X448(const spint *, spint *, size_t, size_t);
void batch_redc_X25519(const spint *, spint *, size_t, size_t);
void batch_redc_NIST256(const spint *, spint *, size_t, size_t);
void batch_redc_X448(const spint *, spint *, size_t, size_t);
void batch_rfc7748_X25519(const char *, const char *, char *, size_t);
void batch_rfc7748_X448(const char *, const char *, char *, size_t);

typedef struct {
    int kind;
    const void *a, *b;
    void *c;
    size_t off, n, ld;
} job_t;

static void *run(void *arg) {
    job_t *j = (job_t *)arg;
    const spint *a = (const spint *)j->a + j->off, *b = (const spint *)j->b + j->off;
    spint *c = (spint *)j->c + j->off;
    switch (j->kind) {
    case 0: batch_modmul_X25519(a, b, c, j->n, j->ld); break;
    case 1: batch_modmul_NIST256(a, b, c, j->n, j->ld); break;
    case 2: batch_modmul_X448(a, b, c, j->n, j->ld); break;
    case 3: batch_rfc7748_X25519((const char *)j->a + j->off * 32, (const char *)j->b + j->off * 32, (char *)j->c + j->off * 32, j->n); break;
    case 4: batch_rfc7748_X448((const char *)j->a + j->off * 56, (const char *)j->b + j->off * 56, (char *)j->c + j->off * 56, j->n); break;
    /* unary ops over SoA: c = op(a), b unused */
    case 5: batch_modsqr_X25519(a, c, j->n, j->ld); break;
    case 6: batch_modsqr_NIST256(a, c, j->n, j->ld); break;
    case 7: batch_modsqr_X448(a, c, j->n, j->ld); break;
    case 8: batch_nres_X25519(a, c, j->n, j->ld); break;
    case 9: batch_nres_NIST256(a, c, j->n, j->ld); break;
    case 10: batch_nres_X448(a, c, j->n, j->ld); break;
    case 11: batch_redc_X25519(a, c, j->n, j->ld); break;
    case 12: batch_redc_NIST256(a, c, j->n, j->ld); break;
    case 13: batch_redc_X448(a, c, j->n, j->ld); break;
    }
    return NULL;
}

/* kind: 0..2 = modmul X25519/NIST256/X448 over SoA (ld = limb stride), 3..4 = rfc7748 X25519/X448,
 * 5..7 = modsqr, 8..10 = nres, 11..13 = redc (X25519/NIST256/X448; b unused) */
int oracle_parallel(int kind, const void *a, const void *b, void *c, size_t n, size_t ld, int threads) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256];
    job_t jobs[256];
    size_t per = (n + (size_t)threads - 1) / (size_t)threads, off = 0;
    int used = 0;
    for (int t = 0; t < threads && off < n; t++) {
        size_t cnt = n - off < per ? n - off : per;
        jobs[t] = (job_t){kind, a, b, c, off, cnt, ld};
        if (pthread_create(&th[t], NULL, run, &jobs[t]) != 0) return -1;
        off += cnt;
        used++;
    }
    for (int t = 0; t < used; t++) pthread_join(th[t], NULL);
    return 0;
}
