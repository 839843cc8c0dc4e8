/* examples/rfc7748_drop_in.c -- the reference's rfc7748.c main() sequence (rfc7748.c:259-341) against
 * libmodarith_amd.so instead of a pasted field.c: the RFC 7748 test vector, the 5000 x 2 chained calls
 * keyed by the reference's LCG, and the Diffie-Hellman exchange -- first through the scalar entry point
 * with the reference signature `void rfc7748(const char *bk,const char *bu,char *bv)`, then the same
 * exchange as ONE batched launch.  Plain C, no HIP headers.
 *
 *   gcc -O2 examples/rfc7748_drop_in.c -Iinclude -Lmodarith_amd -l:libmodarith_amd.so \
 *       -Wl,-rpath,$PWD/modarith_amd -o examples/rfc7748_drop_in && examples/rfc7748_drop_in [iterations]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "modarith_amd.h"

#define Nbytes 32
#define GENERATOR 9
#define rfc7748 rfc7748_X25519 /* the only line a consumer of the scalar form changes */

static void to_hex(const char *src, char *dst) {
    static const char *d = "0123456789abcdef";
    for (int i = 0; i < Nbytes; i++) {
        unsigned char ch = (unsigned char)src[i];
        dst[2 * i] = d[ch >> 4];
        dst[2 * i + 1] = d[ch & 15];
    }
    dst[2 * Nbytes] = 0;
}
static void from_hex(const char *src, char *dst) {
    for (int i = 0; i < Nbytes; i++) {
        unsigned v;
        sscanf(src + 2 * i, "%2x", &v);
        dst[i] = (char)v;
    }
}

int main(int argc, char **argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 5000;
    char sv[2 * Nbytes + 1], bk[Nbytes], bv[Nbytes], bu[Nbytes] = {0};
    uint16_t rnd = 1;
    const char *sk = "77076d0a7318a57d3c16c17251b26645df4c2f87ebc0992ab177fba51db92c2a";
    if (modarith_amd_device_count() < 1) { puts("no GPU"); return 2; }

    bu[0] = GENERATOR;
    from_hex(sk, bk);
    rfc7748(bk, bu, bv);
    to_hex(bv, sv);
    printf("Test Vector\n%s\n%s\n", sk, sv);

    memset(bu, 0, Nbytes);
    bu[0] = GENERATOR;
    for (int i = 0; i < Nbytes; i++) { rnd = (uint16_t)(5 * rnd + 1); bk[i] = (char)(rnd % 256); }
    clock_t begin = clock();
    for (int i = 0; i < iters; i++) {
        rfc7748(bk, bu, bv);
        rfc7748(bk, bv, bu);
    }
    printf("Microseconds per call (scalar form, round trip through the GPU)= %d\n",
           (int)(1e6 * (double)(clock() - begin) / CLOCKS_PER_SEC / (2.0 * iters)));
    to_hex(bu, sv);
    printf("chain %d\n%s\n", iters, sv);

    char alice[Nbytes], bob[Nbytes], apk[Nbytes], bpk[Nbytes], ssa[Nbytes], ssb[Nbytes];
    for (int i = 0; i < Nbytes; i++) {
        rnd = (uint16_t)(5 * rnd + 1); alice[i] = (char)(rnd % 256);
        rnd = (uint16_t)(5 * rnd + 1); bob[i] = (char)(rnd % 256);
        apk[i] = bpk[i] = 0;
    }
    apk[0] = bpk[0] = GENERATOR;
    rfc7748(alice, apk, apk);
    rfc7748(bob, bpk, bpk);
    rfc7748(alice, bpk, ssa);
    rfc7748(bob, apk, ssb);
    to_hex(ssa, sv); printf("Alice shared secret\n%s\n", sv);
    to_hex(ssb, sv); printf("Bob's shared secret\n%s\n", sv);

    /* the same two final multiplications as one batched launch of n = 2 records */
    char hk[2 * Nbytes], hu[2 * Nbytes], hv[2 * Nbytes];
    memcpy(hk, alice, Nbytes); memcpy(hk + Nbytes, bob, Nbytes);
    memcpy(hu, bpk, Nbytes);   memcpy(hu + Nbytes, apk, Nbytes);
    void *dk, *du, *dv;
    if (modarith_amd_malloc(&dk, sizeof hk) || modarith_amd_malloc(&du, sizeof hu) || modarith_amd_malloc(&dv, sizeof hv)) return 3;
    modarith_amd_memcpy_h2d(dk, hk, sizeof hk, NULL);
    modarith_amd_memcpy_h2d(du, hu, sizeof hu, NULL);
    if (rfc7748_X25519_batch((const char *)dk, (const char *)du, (char *)dv, 2, NULL)) { puts(modarith_amd_last_error()); return 4; }
    modarith_amd_memcpy_d2h(hv, dv, sizeof hv, NULL);
    modarith_amd_sync(NULL);
    printf("batched: %s\n", (memcmp(hv, ssa, Nbytes) == 0 && memcmp(hv + Nbytes, ssb, Nbytes) == 0) ? "equal" : "DIFFERENT");
    modarith_amd_free(dk); modarith_amd_free(du); modarith_amd_free(dv);
    return 0;
}
