/* examples/ecdsa_verify_batch.c -- NIST256_VERIFY of the reference (nist256.c:226-260) for n signatures at once, in plain C
 * against libmodarith_amd.so: every gel / point of that function becomes a device batch, every call keeps its name with _batch
 * appended, the group-order arithmetic runs on the NIST256Q field (curve.py:324-329), and ecnXXXgen + ecnXXXmul2 + ecnXXXget is the
 * fused verification kernel.  Input: the FIPS 186 P-256 / SHA-256 signature vector (the key and message hash of nist256.c:266-268)
 * in every lane, with three lanes of four made invalid in the three ways the reference rejects -- another message, s = 0, r out of
 * range -- so the expected verdicts are 1 0 0 0 1 0 0 0 ...
 *
 *   gcc -O2 examples/ecdsa_verify_batch.c -Iinclude -Lmodarith_amd -l:libmodarith_amd.so \
 *       -Wl,-rpath,$PWD/modarith_amd -o examples/ecdsa_verify_batch && examples/ecdsa_verify_batch [n]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "modarith_amd.h"

#define BYTES 32
#define NL 5 /* limbs of a NIST256 / NIST256Q element */
#define CHECK(call) do { int rc_ = (call); if (rc_) { printf("%s failed: %s\n", #call, modarith_amd_last_error()); return 1; } } while (0)

static void from_hex(const char *src, char *dst) {
    for (int i = 0; i < BYTES; i++) { unsigned v; sscanf(src + 2 * i, "%2x", &v); dst[i] = (char)v; }
}
static int dev_bytes(void **d, const char *h, size_t bytes) {
    if (modarith_amd_malloc(d, bytes)) return 1;
    return h ? modarith_amd_memcpy_h2d(*d, h, bytes, NULL) : 0;
}

int main(int argc, char **argv) {
    size_t n = argc > 1 ? (size_t)atol(argv[1]) : 4096;
    char qx[BYTES], qy[BYTES], e[BYTES], r[BYTES], s[BYTES], e_bad[BYTES], zero[BYTES], big[BYTES];
    from_hex("1ccbe91c075fc7f4f033bfa248db8fccd3565de94bbfb12f3c59ff46c271bf83", qx);
    from_hex("ce4014c68811f9a21a1fdb2c0e6113e06db7ca93b7404e78dc7ccd5ca89a4ca9", qy);
    from_hex("44acf6b7e36c1342c2c5897204fe09504e1e2efb1a900377dbc4e7a6a133ec56", e);
    from_hex("f3ac8061b514795b8843e3d6629527ed2afd6b1f6a555a7acabb5e6f79c8c2ac", r);
    from_hex("8bf77819ca05a6b2786c76262bf7371cef97b218e96f175a3ccdda2acc058903", s);
    memcpy(e_bad, e, BYTES); e_bad[31] ^= 1;
    memset(zero, 0, BYTES);
    memset(big, 0xff, BYTES);                                  /* >= q: modimp reports it out of range (nist256.c:240-241) */

    char *hx = malloc(n * BYTES), *hy = malloc(n * BYTES), *he = malloc(n * BYTES), *hr = malloc(n * BYTES), *hs = malloc(n * BYTES);
    for (size_t j = 0; j < n; j++) {
        memcpy(hx + j * BYTES, qx, BYTES);
        memcpy(hy + j * BYTES, qy, BYTES);
        memcpy(he + j * BYTES, j % 4 == 1 ? e_bad : e, BYTES);
        memcpy(hr + j * BYTES, j % 4 == 3 ? big : r, BYTES);
        memcpy(hs + j * BYTES, j % 4 == 2 ? zero : s, BYTES);
    }
    void *dx, *dy, *de, *dr, *ds, *du, *dv, *drb;
    if (dev_bytes(&dx, hx, n * BYTES) || dev_bytes(&dy, hy, n * BYTES) || dev_bytes(&de, he, n * BYTES) || dev_bytes(&dr, hr, n * BYTES) ||
        dev_bytes(&ds, hs, n * BYTES) || dev_bytes(&du, NULL, n * BYTES) || dev_bytes(&dv, NULL, n * BYTES) || dev_bytes(&drb, NULL, n * BYTES)) {
        printf("device memory: %s\n", modarith_amd_last_error());
        return 1;
    }
    /* gel e, r, s, rds: batches of NL limbs, limb-major (element j of limb i at [i * n + j]: ld = n) */
    ma_spint *E, *R, *S, *T;
    int *r_ok, *s_ok, *r_0, *s_0, *same;
    const size_t gel = NL * n * sizeof(ma_spint);
    CHECK(modarith_amd_malloc((void **)&E, gel)); CHECK(modarith_amd_malloc((void **)&R, gel));
    CHECK(modarith_amd_malloc((void **)&S, gel)); CHECK(modarith_amd_malloc((void **)&T, gel));
    CHECK(modarith_amd_malloc((void **)&r_ok, 5 * n * sizeof(int)));
    s_ok = r_ok + n; r_0 = s_ok + n; s_0 = r_0 + n; same = s_0 + n;
    ma_spint *Q;
    CHECK(modarith_amd_malloc((void **)&Q, 3 * gel));
    size_t wsb = ecn_nist256_mulgen2_get_workspace_bytes(n);
    void *ws;
    CHECK(modarith_amd_malloc(&ws, wsb));

    CHECK(modimp_NIST256Q_batch((const char *)de, E, same, n, n, NULL));          /* modimp(thm,e);  (flag unused, as in the reference) */
    CHECK(modimp_NIST256Q_batch((const char *)dr, R, r_ok, n, n, NULL));          /* if (!modimp(sig,r)) return 0;                       */
    CHECK(modimp_NIST256Q_batch((const char *)ds, S, s_ok, n, n, NULL));          /* if (!modimp(&sig[BYTES],s)) return 0;               */
    CHECK(modis0_NIST256Q_batch(R, r_0, n, n, NULL));                             /* if (modis0(r) || modis0(s)) return 0;               */
    CHECK(modis0_NIST256Q_batch(S, s_0, n, n, NULL));
    CHECK(modinv_NIST256Q_batch(S, NULL, S, n, n, NULL));                         /* modinv(s,NULL,s);   (one inversion per 64 lanes)    */
    CHECK(modmul_NIST256Q_batch(R, S, T, n, n, NULL));                            /* modmul(r,s,rds); modexp(rds,v);                     */
    CHECK(modexp_NIST256Q_batch(T, (char *)dv, n, n, NULL));
    CHECK(modmul_NIST256Q_batch(S, E, S, n, n, NULL));                            /* modmul(s,e,s); modexp(s,u);                         */
    CHECK(modexp_NIST256Q_batch(S, (char *)du, n, n, NULL));
    CHECK(ecn_nist256_set_batch(NULL, (const char *)dx, (const char *)dy, Q, n, n, NULL));      /* ecnXXXset(0,&pub[1],&pub[BYTES+1],&Q) */
    /* ecnXXXgen(&G); ecnXXXmul2(u,&G,v,&Q,&Q); ecnXXXget(&Q,rb,NULL): one kernel; infinity leaves as (0, 1) */
    CHECK(ecn_nist256_mulgen2_get_batch((const char *)du, (const char *)dv, Q, (char *)drb, (char *)dy, NULL, n, n, ws, wsb, NULL));
    CHECK(modimp_NIST256Q_batch((const char *)drb, E, same, n, n, NULL));         /* modimp(rb,e);                                       */
    CHECK(modcmp_NIST256Q_batch(R, E, same, n, n, NULL));                         /* if (modcmp(r,e)) return 1;                          */

    int *flags = malloc(5 * n * sizeof(int));
    CHECK(modarith_amd_memcpy_d2h(flags, r_ok, 5 * n * sizeof(int), NULL));
    CHECK(modarith_amd_memcpy_d2h(hx, drb, n * BYTES, NULL));
    CHECK(modarith_amd_memcpy_d2h(hy, dy, n * BYTES, NULL));
    CHECK(modarith_amd_sync(NULL));
    size_t valid = 0, as_expected = 0;
    for (size_t j = 0; j < n; j++) {
        int inf = hy[j * BYTES + BYTES - 1] == 1;                                 /* if (ecnXXXisinf(&Q)) return 0;  x = 0, y = 1          */
        for (int i = 0; i < BYTES; i++) inf &= hx[j * BYTES + i] == 0 && (i == BYTES - 1 || hy[j * BYTES + i] == 0);
        int ok = flags[j] && flags[n + j] && !flags[2 * n + j] && !flags[3 * n + j] && !inf && flags[4 * n + j];
        valid += ok;
        as_expected += ok == (j % 4 == 0);
    }
    printf("%zu signatures, %zu valid, %zu verdicts as expected\n", n, valid, as_expected);
    printf("verification: %s\n", as_expected == n ? "as the reference decides" : "DIFFERENT");
    modarith_amd_free(dx); modarith_amd_free(dy); modarith_amd_free(de); modarith_amd_free(dr); modarith_amd_free(ds);
    modarith_amd_free(du); modarith_amd_free(dv); modarith_amd_free(drb); modarith_amd_free(E); modarith_amd_free(R);
    modarith_amd_free(S); modarith_amd_free(T); modarith_amd_free(r_ok); modarith_amd_free(Q); modarith_amd_free(ws);
    free(hx); free(hy); free(he); free(hr); free(hs); free(flags);
    return as_expected == n ? 0 : 1;
}
