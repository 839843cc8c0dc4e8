/* examples/generated_field.c -- a C consumer of a field made by the generator mode.  The reference user with a prime of
 * their own runs `python pseudo.py 64 2**251-9` and pastes the emitted field.c; here
 *
 *   python -m modarith_amd.generate 64 2**251-9        (tag 2519, the generators' own decoration for an unnamed 2^251 - 9)
 *   gcc -O2 examples/generated_field.c -Iinclude -Lmodarith_amd/plugins -l:libmodarith_amd_2519.so -Lmodarith_amd \
 *       -l:libmodarith_amd.so -Wl,-rpath,$PWD/modarith_amd/plugins -Wl,-rpath,$PWD/modarith_amd -o examples/generated_field
 *
 * and the same function names are there, scalar (host pointers, reference signatures) and batched (device pointers).
 * The program runs the generators' acceptance chain (pseudo.py:1783-1796): z = 1 / ((x - y)(x + y))^2, checks z * that = 1
 * through the scalar entry points, then does the chain batched on the device and compares every element with the scalar path.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "modarith_amd.h"

MODARITH_AMD_DECLARE(2519)

#define Nlimbs 5
#define CK(call) do { if ((call) != 0) { printf("%s failed: %s\n", #call, modarith_amd_last_error()); return 1; } } while (0)

static uint64_t rs = 0x9e3779b97f4a7c15ull;
static uint64_t rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }

static void chain_scalar(const ma_spint *x0, const ma_spint *y0, ma_spint *z, ma_spint *s) {
    ma_spint x[Nlimbs], y[Nlimbs], t[Nlimbs];
    nres_2519_ct(x0, x); nres_2519_ct(y0, y);
    modadd_2519_ct(x, y, t); modsub_2519_ct(x, y, s);
    modmul_2519_ct(t, s, s); modsqr_2519_ct(s, s);
    modinv_2519_ct(s, NULL, z);
}

int main(int argc, char **argv) {
    size_t n = argc > 1 ? (size_t)atol(argv[1]) : 1000;
    if (modarith_amd_device_count() < 1) { puts("no GPU"); return 2; }
    ma_spint *x = malloc(n * Nlimbs * 8), *y = malloc(n * Nlimbs * 8), *z = malloc(n * Nlimbs * 8);   /* limb-interleaved: x[i*n + j] */
    for (size_t j = 0; j < n; j++)
        for (int i = 0; i < Nlimbs; i++) { x[i * n + j] = rnd() & ((1ull << 51) - 1); y[i * n + j] = rnd() & ((1ull << 51) - 1); }

    /* scalar: z * ((x-y)(x+y))^2 == 1 */
    ma_spint a[Nlimbs], b[Nlimbs], zs[Nlimbs], s[Nlimbs], one[Nlimbs];
    for (int i = 0; i < Nlimbs; i++) { a[i] = x[i * n]; b[i] = y[i * n]; }
    chain_scalar(a, b, zs, s);
    modmul_2519_ct(zs, s, one); redc_2519_ct(one, one);
    if (!(one[0] == 1 && one[1] == 0 && one[2] == 0 && one[3] == 0 && one[4] == 0)) { puts("scalar chain: z * s != 1"); return 1; }

    /* batched on the device */
    void *dx, *dy, *dt, *dz;
    size_t bytes = n * Nlimbs * 8;
    CK(modarith_amd_malloc(&dx, bytes)); CK(modarith_amd_malloc(&dy, bytes)); CK(modarith_amd_malloc(&dt, bytes)); CK(modarith_amd_malloc(&dz, bytes));
    CK(modarith_amd_memcpy_h2d(dx, x, bytes, NULL)); CK(modarith_amd_memcpy_h2d(dy, y, bytes, NULL));
    CK(nres_2519_batch(dx, dx, n, n, NULL)); CK(nres_2519_batch(dy, dy, n, n, NULL));
    CK(modadd_2519_batch(dx, dy, dt, n, n, NULL)); CK(modsub_2519_batch(dx, dy, dz, n, n, NULL));
    CK(modmul_2519_batch(dt, dz, dz, n, n, NULL)); CK(modsqr_2519_batch(dz, dz, n, n, NULL));
    CK(modinv_2519_batch(dz, NULL, dz, n, n, NULL));
    CK(modarith_amd_memcpy_d2h(z, dz, bytes, NULL)); CK(modarith_amd_sync(NULL));
    size_t bad = 0;
    for (size_t j = 0; j < n; j++) {
        for (int i = 0; i < Nlimbs; i++) { a[i] = x[i * n + j]; b[i] = y[i * n + j]; }
        chain_scalar(a, b, zs, s);
        for (int i = 0; i < Nlimbs; i++) bad += zs[i] != z[i * n + j];
        if (j >= 64 && j + 64 < n) j += n / 97;            /* every element at both ends, a stride in between */
    }
    modarith_amd_free(dx); modarith_amd_free(dy); modarith_amd_free(dt); modarith_amd_free(dz);
    if (bad) { printf("batched chain differs from the scalar entry points in %zu limbs\n", bad); return 1; }
    printf("2^251-9 (generated, tag 2519): scalar chain inverts, batched chain over %zu elements equal to the scalar entry points\n", n);
    return 0;
}
