/* oracle/time_oracle.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 * The reference's time.c protocol (pseudo.py:1177-1386, main 1448-1459; monty.py:1650-1857) on the
 * oracle field code: seed-42 operands (random.seed(42), four randint(0,p-1), pseudo.py:1862-1866)
 * baked in as constants, 10^8 dependent modmul, 10^8 modsqr, 10^5 modinv, single thread; prints the
 * 24-bit check words, which must equal tests/golden/field_<PRIME>.json "time".
 *   usage: time_oracle [scale]    (scale divides the loop counts like the generators' `scale`) */
#include "oracle_types.h"
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#define DECL(P)                                                         \
    unsigned int time_modmul_##P(spint *, spint *, long);               \
    unsigned int time_modsqr_##P(spint *, long);                        \
    unsigned int time_modinv_##P(spint *, long);
DECL(X25519) DECL(NIST256) DECL(X448)

static double now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* makebig (pseudo.py:190-199): hex big-endian string -> n limbs of `radix` bits, all masked */
static void limbs_of(const char *hex, int radix, int n, spint *out) {
    unsigned __int128 acc = 0;
    int bits = 0, k = 0, len = 0;
    while (hex[len]) len++;
    for (int i = 0; i < n; i++) out[i] = 0;
    for (int i = len - 1; i >= 0; i--) {
        char ch = hex[i];
        unsigned d = ch <= '9' ? (unsigned)(ch - '0') : (unsigned)((ch | 32) - 'a' + 10);
        acc |= (unsigned __int128)d << bits;
        bits += 4;
        while (bits >= radix && k < n) {
            out[k++] = (spint)(acc & ((((unsigned __int128)1) << radix) - 1));
            acc >>= radix;
            bits -= radix;
        }
    }
    if (k < n) out[k] = (spint)acc;
}

#define RUN(P, RADIX_, NL_, RA, RB, RS, RI)                                                     \
    do {                                                                                         \
        spint x[NL_], y[NL_];                                                                    \
        double t0;                                                                               \
        unsigned int w;                                                                          \
        limbs_of(RA, RADIX_, NL_, x); limbs_of(RB, RADIX_, NL_, y);                              \
        t0 = now(); w = time_modmul_##P(x, y, 100000 / scale);                                   \
        printf(#P " modmul check 0x%06x ns/op %.2f\n", w, (now() - t0) * 1e9 / (1e8 / scale));   \
        limbs_of(RS, RADIX_, NL_, x);                                                            \
        t0 = now(); w = time_modsqr_##P(x, 100000 / scale);                                      \
        printf(#P " modsqr check 0x%06x ns/op %.2f\n", w, (now() - t0) * 1e9 / (1e8 / scale));   \
        limbs_of(RI, RADIX_, NL_, x);                                                            \
        t0 = now(); w = time_modinv_##P(x, 50000 / scale);                                       \
        printf(#P " modinv check 0x%06x ns/op %.2f\n", w, (now() - t0) * 1e9 / (1e5 / scale));   \
    } while (0)

int main(int argc, char **argv) {
    long scale = argc > 1 ? atol(argv[1]) : 1;
    if (scale < 1) scale = 1;
    RUN(X25519, 51, 5,
        "11dc60f4392456de3eb13b9046685257bdd640fb06671ad11c80317fa3b1799d",
        "4b95423416419f828b9d2434e465e150bd9c66b3ad3c2d6d1a3d1fa7bc8960a9",
        "4d0ef322815ef6d13b8faa1837f8a88b17fc695a07a0ca6e0822e8f36c031199",
        "35b2d3528b8148f6b38a088ca65ed389b74d0fb132e706298fadc1a606cb0fb3");
    RUN(NIST256, 52, 5,
        "23b8c1e9392456de3eb13b9046685257bdd640fb06671ad11c80317fa3b1799d",
        "972a846916419f828b9d2434e465e150bd9c66b3ad3c2d6d1a3d1fa7bc8960a9",
        "9a1de644815ef6d13b8faa1837f8a88b17fc695a07a0ca6e0822e8f36c031199",
        "6b65a6a48b8148f6b38a088ca65ed389b74d0fb132e706298fadc1a606cb0fb3");
    RUN(X448, 56, 8,
        "8b9d2434e465e150bd9c66b3ad3c2d6d1a3d1fa7bc8960a923b8c1e9392456de3eb13b9046685257bdd640fb06671ad11c80317fa3b1799d",
        "b74d0fb132e706298fadc1a606cb0fb39a1de644815ef6d13b8faa1837f8a88b17fc695a07a0ca6e0822e8f36c031199972a846916419f82",
        "28df6ec4ce4a2bbdc241330b01a9e71fde8a774bcf36d58b4737819096da1dac72ff5d2a386ecbe06b65a6a48b8148f6b38a088ca65ed389",
        "5be6128e18c267976142ea7d17be31111a2a73ed562b0f79c37459eef50bea63371ecd7b27cd813047229389571aa8766c307511b2b9437a");
    return 0;
}
