/* examples/ecn_batch.c -- the reference's curve API (curve.h:13-29) against libmodarith_amd.so, scalar and batched, on
 * secp256k1: the check of testcurve.c:224-237 (order*G = O; r1*G + r2*G = O through ecnXXXmul2) with the reference's
 * own function names, then n public keys k_j*G in ONE batched launch, compared with the scalar path, then the fused forms of the
 * key-generation / verification call sequences against those.
 * Plain C, no HIP headers.
 *
 *   gcc -O2 examples/ecn_batch.c -Iinclude -Lmodarith_amd -l:libmodarith_amd.so \
 *       -Wl,-rpath,$PWD/modarith_amd -o examples/ecn_batch && examples/ecn_batch [n]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "modarith_amd.h"

#define Nbytes 32
#define Nlimbs 5
/* what curve.py's substitution XXX -> _secp256k1_ produces; a consumer of curve.h changes nothing else */
typedef ma_point_secp256k1_t point;
#define ecnXXXgen ecn_secp256k1_gen
#define ecnXXXmul ecn_secp256k1_mul
#define ecnXXXmul2 ecn_secp256k1_mul2
#define ecnXXXisinf ecn_secp256k1_isinf
#define ecnXXXget ecn_secp256k1_get
#define ecnXXXcpy ecn_secp256k1_cpy

static void from_hex(const char *src, char *dst) {
    for (int i = 0; i < Nbytes; i++) {
        unsigned v;
        sscanf(src + 2 * i, "%2x", &v);
        dst[i] = (char)v;
    }
}
static void print_hex(const char *b) {
    for (int i = 0; i < Nbytes; i++) printf("%02x", (unsigned char)b[i]);
    printf("\n");
}
#define CHECK(call) do { int rc_ = (call); if (rc_) { printf("%s failed: %s\n", #call, modarith_amd_last_error()); return 1; } } while (0)

int main(int argc, char **argv) {
    size_t n = argc > 1 ? (size_t)atol(argv[1]) : 4096;
    /* testcurve.c:78-84 */
    char order[Nbytes], r1[Nbytes], r2[Nbytes], x[Nbytes], y[Nbytes];
    from_hex("FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141", order);
    from_hex("166876CB6C86C76660666789A376F6790956A0D6A507657196D75D610E0C9D7B", r1);
    from_hex("E9978934937938999F9998765C890985B1583C100A413ACA28FB012BC229A3C6", r2);
    point G, P, Q;
    ecnXXXgen(&G);
    ecnXXXget(&G, x, y);
    printf("generator x\n"); print_hex(x);
    ecnXXXcpy(&G, &P);
    ecnXXXmul(order, &P);
    printf("order*G is the neutral element: %s\n", ecnXXXisinf(&P) ? "yes" : "NO");
    ecnXXXcpy(&G, &P);
    ecnXXXmul2(r1, &P, r2, &P, &Q);
    printf("r1*G + r2*G is the neutral element: %s\n", ecnXXXisinf(&Q) ? "yes" : "NO");

    /* n public keys in one launch: device SoA points P[(c*Nlimbs + i)*n + j], big-endian scalar records */
    char *he = malloc(n * Nbytes), *hx = malloc(n * Nbytes);
    uint64_t s = 42;
    for (size_t i = 0; i < n * Nbytes; i++) { s = s * 6364136223846793005ull + 1442695040888963407ull; he[i] = (char)(s >> 56); }
    void *dP, *de, *dx, *ws;
    size_t wsb = ecn_secp256k1_mul_workspace_bytes(n);
    CHECK(modarith_amd_malloc(&dP, 3 * Nlimbs * n * sizeof(ma_spint)));
    CHECK(modarith_amd_malloc(&de, n * Nbytes));
    CHECK(modarith_amd_malloc(&dx, n * Nbytes));
    CHECK(modarith_amd_malloc(&ws, wsb));
    CHECK(modarith_amd_memcpy_h2d(de, he, n * Nbytes, NULL));
    CHECK(ecn_secp256k1_gen_batch((ma_spint *)dP, n, n, NULL));
    CHECK(ecn_secp256k1_mul_batch((const char *)de, (ma_spint *)dP, n, n, ws, wsb, NULL));
    CHECK(ecn_secp256k1_get_batch((ma_spint *)dP, (char *)dx, NULL, NULL, n, n, NULL));
    CHECK(modarith_amd_memcpy_d2h(hx, dx, n * Nbytes, NULL));
    CHECK(modarith_amd_sync(NULL));
    /* the same for a few of them through the scalar API */
    int equal = 1;
    for (size_t j = 0; j < n; j += (n > 8 ? n / 8 : 1)) {
        ecnXXXgen(&P);
        ecnXXXmul(he + j * Nbytes, &P);
        ecnXXXget(&P, x, NULL);
        equal &= memcmp(x, hx + j * Nbytes, Nbytes) == 0;
    }
    printf("public key 0 x\n"); print_hex(hx);

    /* the same call sequences fused into one kernel each (byte-identical results):
     *   ecnXXXgen + ecnXXXmul + ecnXXXget            -> ecn_secp256k1_mulgen_get_batch   (key generation / signing)
     *   ecnXXXmul + ecnXXXget on a given point       -> ecn_secp256k1_mul_get_batch
     *   ecnXXXgen + ecnXXXmul2(e, G, f, Q) + ecnXXXget -> ecn_secp256k1_mulgen2_get_batch  (verification)                */
    char *hf = malloc(n * Nbytes);
    int fused = 1;
    void *dx2, *dy2, *dQ, *ws2;
    CHECK(modarith_amd_malloc(&dx2, n * Nbytes));
    CHECK(modarith_amd_malloc(&dy2, n * Nbytes));
    CHECK(modarith_amd_malloc(&dQ, 3 * Nlimbs * n * sizeof(ma_spint)));
    CHECK(ecn_secp256k1_mulgen_get_batch((const char *)de, (char *)dx2, NULL, NULL, n, NULL));
    CHECK(modarith_amd_memcpy_d2h(hf, dx2, n * Nbytes, NULL));
    CHECK(modarith_amd_sync(NULL));
    fused &= memcmp(hf, hx, n * Nbytes) == 0;
    size_t wsb2 = ecn_secp256k1_mul_get_workspace_bytes(n), wsb3 = ecn_secp256k1_mulgen2_get_workspace_bytes(n);
    CHECK(modarith_amd_malloc(&ws2, wsb2 > wsb3 ? wsb2 : wsb3));
    CHECK(ecn_secp256k1_gen_batch((ma_spint *)dQ, n, n, NULL));
    CHECK(ecn_secp256k1_mul_get_batch((const char *)de, (const ma_spint *)dQ, (char *)dx2, (char *)dy2, NULL, n, n, ws2, wsb2, NULL));
    CHECK(modarith_amd_memcpy_d2h(hf, dx2, n * Nbytes, NULL));
    CHECK(modarith_amd_sync(NULL));
    fused &= memcmp(hf, hx, n * Nbytes) == 0;
    /* verification shape: e*G + e*Q with Q = e*G left in dP by the batched mul above (projective, not normalised) */
    CHECK(ecn_secp256k1_mulgen2_get_batch((const char *)de, (const char *)de, (const ma_spint *)dP, (char *)dx2, NULL, NULL, n, n, ws2, wsb3, NULL));
    CHECK(modarith_amd_memcpy_d2h(hf, dx2, n * Nbytes, NULL));
    CHECK(modarith_amd_sync(NULL));
    for (size_t j = 0; j < n; j += (n > 4 ? n / 4 : 1)) {
        point A, B, R;
        ecnXXXgen(&A);
        ecnXXXgen(&B);
        ecnXXXmul(he + j * Nbytes, &B);
        ecnXXXmul2(he + j * Nbytes, &A, he + j * Nbytes, &B, &R);
        ecnXXXget(&R, x, NULL);
        fused &= memcmp(x, hf + j * Nbytes, Nbytes) == 0;
    }
    printf("fused == call sequences: %s\n", fused ? "equal" : "DIFFERENT");
    printf("batched == scalar: %s\n", equal ? "equal" : "DIFFERENT");
    modarith_amd_free(dP); modarith_amd_free(de); modarith_amd_free(dx); modarith_amd_free(ws);
    modarith_amd_free(dx2); modarith_amd_free(dy2); modarith_amd_free(dQ); modarith_amd_free(ws2);
    free(he); free(hx); free(hf);
    return (equal && fused) ? 0 : 1;
}
