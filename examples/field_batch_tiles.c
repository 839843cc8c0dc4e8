/* examples/field_batch_tiles.c -- the batched field API from plain C, as a consumer that holds its elements the way field.c's
 * callers do (spint x[n][Nlimbs], element-major on the host) would use it: upload, turn the element-major arrays into the
 * tiled limb-interleaved layout on the device (ld = 4096 selects tiles), run the reference's own acceptance chain
 * (pseudo.py:1783-1796: nres, nres, modadd, modsub, modmul, modsqr, modinv, ... redc) over the whole batch, bring the
 * results back element-major, and check a few of them against the scalar entry points with the reference signatures
 * (modmul_X25519_ct etc.).  No HIP headers, no C++.
 *
 *   gcc -O2 examples/field_batch_tiles.c -Iinclude -Lmodarith_amd -l:libmodarith_amd.so \
 *       -Wl,-rpath,$PWD/modarith_amd -o examples/field_batch_tiles && examples/field_batch_tiles [n]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "modarith_amd.h"

#define Nlimbs 5
#define CK(call) do { if ((call) != 0) { printf("%s failed: %s\n", #call, modarith_amd_last_error()); return 1; } } while (0)

static uint64_t rnd_state = 88172645463325252ull;
static uint64_t rnd(void) { rnd_state ^= rnd_state << 13; rnd_state ^= rnd_state >> 7; rnd_state ^= rnd_state << 17; return rnd_state; }

int main(int argc, char **argv) {
    size_t n = argc > 1 ? (size_t)atol(argv[1]) : (size_t)5 * 4096 + 1237;      /* odd, with a partial last tile */
    /* the layout is the library's recommendation for this batch size: tiles of 4096 from two whole tiles on (flat rows
     * below), and the buffer size that goes with it -- a caller never hard-codes either */
    const size_t TILE = modarith_amd_recommended_ld(n), words = modarith_amd_batch_words(n, Nlimbs, TILE);
    if (modarith_amd_device_count() < 1) { puts("no GPU"); return 2; }
    ma_spint (*x)[Nlimbs] = malloc(n * sizeof *x), (*y)[Nlimbs] = malloc(n * sizeof *y), (*z)[Nlimbs] = malloc(n * sizeof *z);
    for (size_t j = 0; j < n; j++)
        for (int i = 0; i < Nlimbs; i++) { x[j][i] = rnd() & ((1ull << 51) - 1); y[j][i] = rnd() & ((1ull << 51) - 1); }

    void *dx_aos, *dy_aos, *dx, *dy, *dt, *dz;                     /* device: two element-major arrays, four tiled batches */
    CK(modarith_amd_malloc(&dx_aos, n * sizeof *x)); CK(modarith_amd_malloc(&dy_aos, n * sizeof *y));
    CK(modarith_amd_malloc(&dx, words * 8)); CK(modarith_amd_malloc(&dy, words * 8));
    CK(modarith_amd_malloc(&dt, words * 8)); CK(modarith_amd_malloc(&dz, words * 8));
    CK(modarith_amd_memcpy_h2d(dx_aos, x, n * sizeof *x, NULL)); CK(modarith_amd_memcpy_h2d(dy_aos, y, n * sizeof *y, NULL));
    CK(modarith_amd_aos_to_soa(dx_aos, dx, n, Nlimbs, TILE, NULL));          /* ld = TILE < n: the tiled layout */
    CK(modarith_amd_aos_to_soa(dy_aos, dy, n, Nlimbs, TILE, NULL));

    /* the generators' acceptance chain, batched: z = 1 / ((x - y)(x + y))^2 */
    CK(nres_X25519_batch(dx, dx, n, TILE, NULL)); CK(nres_X25519_batch(dy, dy, n, TILE, NULL));
    CK(modadd_X25519_batch(dx, dy, dt, n, TILE, NULL));
    CK(modsub_X25519_batch(dx, dy, dz, n, TILE, NULL));
    CK(modmul_X25519_batch(dt, dz, dz, n, TILE, NULL));
    CK(modsqr_X25519_batch(dz, dz, n, TILE, NULL));
    CK(modinv_X25519_batch(dz, NULL, dz, n, TILE, NULL));                   /* in place; one inversion per 64 elements */
    CK(redc_X25519_batch(dz, dz, n, TILE, NULL));
    CK(modarith_amd_soa_to_aos(dz, dx_aos, n, Nlimbs, TILE, NULL));
    CK(modarith_amd_memcpy_d2h(z, dx_aos, n * sizeof *z, NULL));
    CK(modarith_amd_sync(NULL));

    /* the same chain through the scalar entry points (reference signatures) on a few elements */
    size_t probe[] = {0, 1, TILE - 1, TILE, n / 2, n - 2, n - 1};
    int bad = 0;
    for (size_t k = 0; k < sizeof probe / sizeof probe[0]; k++) {
        size_t j = probe[k] < n ? probe[k] : n - 1;
        ma_spint a[Nlimbs], b[Nlimbs], t[Nlimbs], w[Nlimbs];
        memcpy(a, x[j], sizeof a); memcpy(b, y[j], sizeof b);
        nres_X25519_ct(a, a); nres_X25519_ct(b, b);
        modadd_X25519_ct(a, b, t); modsub_X25519_ct(a, b, w);
        modmul_X25519_ct(t, w, w); modsqr_X25519_ct(w, w);
        modinv_X25519_ct(w, NULL, w); redc_X25519_ct(w, w);
        if (memcmp(w, z[j], sizeof w) != 0) { bad++; printf("element %zu differs\n", j); }
    }
    printf("batched chain over %zu elements, limb stride %zu (%s): %s\n", n, TILE, TILE < n ? "tiles" : "flat rows", bad ? "MISMATCH" : "equal to the scalar entry points");
    modarith_amd_free(dx_aos); modarith_amd_free(dy_aos); modarith_amd_free(dx); modarith_amd_free(dy); modarith_amd_free(dt); modarith_amd_free(dz);
    free(x); free(y); free(z);
    return bad != 0;
}
