/* examples/fused_chain.c -- a fused chain from plain C.  The chain is built once, outside the program,
 *
 *   python -m modarith_amd.fuse X25519 bench_prod "in a, b; out modsqr(modmul(modadd(a, b), modsub(a, b)))"
 *
 * (it is also one of the plug-ins __graft_entry__.build() makes) and is then an ordinary C function over device batches:
 * z = ((a + b)(a - b))^2 in ONE kernel, 120 bytes per element instead of the 440 of the four calls.  The program runs it
 * next to the four batched calls and compares every limb.
 *
 *   gcc -O2 examples/fused_chain.c -Iinclude -Lmodarith_amd/plugins -l:libmodarith_amd_chain_bench_prod_X25519.so -Lmodarith_amd \
 *       -l:libmodarith_amd.so -Wl,-rpath,$PWD/modarith_amd/plugins -Wl,-rpath,$PWD/modarith_amd -o examples/fused_chain
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "modarith_amd.h"

int chain_bench_prod_X25519_batch(const void *const *in, void *const *out, size_t n, size_t ld, void *stream);

#define Nlimbs 5
#define CK(call) do { if ((call) != 0) { printf("%s failed: %s\n", #call, modarith_amd_last_error()); return 1; } } while (0)

int main(int argc, char **argv) {
    size_t n = argc > 1 ? (size_t)atol(argv[1]) : 100000, bytes = n * Nlimbs * 8;
    if (modarith_amd_device_count() < 1) { puts("no GPU"); return 2; }
    void *a, *b, *t, *w, *z1, *z2;
    CK(modarith_amd_malloc(&a, bytes)); CK(modarith_amd_malloc(&b, bytes)); CK(modarith_amd_malloc(&t, bytes));
    CK(modarith_amd_malloc(&w, bytes)); CK(modarith_amd_malloc(&z1, bytes)); CK(modarith_amd_malloc(&z2, bytes));
    CK(moduniform_X25519_batch(42, 0, 0, 0, a, n, n, NULL));             /* the benchmark's input recipe, generated on the device */
    CK(moduniform_X25519_batch(42, 1, 0, 0, b, n, n, NULL));
    /* call by call */
    CK(modadd_X25519_batch(a, b, t, n, n, NULL)); CK(modsub_X25519_batch(a, b, w, n, n, NULL));
    CK(modmul_X25519_batch(t, w, t, n, n, NULL)); CK(modsqr_X25519_batch(t, z1, n, n, NULL));
    /* fused */
    const void *in[2] = {a, b};
    void *out[1] = {z2};
    CK(chain_bench_prod_X25519_batch(in, out, n, n, NULL));
    uint64_t *h1 = malloc(bytes), *h2 = malloc(bytes);
    CK(modarith_amd_memcpy_d2h(h1, z1, bytes, NULL)); CK(modarith_amd_memcpy_d2h(h2, z2, bytes, NULL)); CK(modarith_amd_sync(NULL));
    if (memcmp(h1, h2, bytes) != 0) { puts("fused chain differs from the four calls"); return 1; }
    printf("fused chain over %zu elements: equal to the four calls, limb for limb\n", n);
    return 0;
}
