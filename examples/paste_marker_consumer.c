/* examples/paste_marker_consumer.c -- a consumer of the PASTE-MARKER boundary, compiled and run by tests/test_gpu_paste_marker.py.
 *
 * The reference's templates take their field arithmetic by textual inclusion: "paste field.c here" (rfc7748.c:24-28,
 * edwards.c:19-23, weierstrass.c:16-20, edge.c:5-9).  What they need from the pasted text is the macro block (spint, Nlimbs, Nbytes,
 * Nbits, Radix, Wordlength, MERSENNE | MONTGOMERY, MULBYINT, the curve selector) and the undecorated function names
 * (pseudo.py:1388-1445).  include/field_<PRIME>.h provides exactly that on top of libmodarith_amd.so.
 *
 * This file is written against those macros and undecorated names ONLY -- no modarith_amd_* call, no _ct / _batch name, no
 * prime-specific constant that the macros do not give.  It is its own program (RFC 7748 section 5 as written there: decode, the
 * ladder with a24, x2 * z2^(p-2), encode), not the reference's rfc7748.c; it makes the calls SURVEY Appendix B lists for that file
 * -- modimp modcpy modone modzer modcsw modadd modsub modsqr modmul modmli modpro modinv modexp -- on Nlimbs-sized arrays, with the
 * in / out aliasing the reference uses (modmul(z2, E, z2), modsqr(a, a), modinv(z, h, z)).  Every call is one element through the
 * GPU (the scalar form is a bring-up path: tens of microseconds per call).
 *
 *   gcc -O2 [-DUSE_X448] examples/paste_marker_consumer.c -Iinclude -Lmodarith_amd -l:libmodarith_amd.so \
 *       -Wl,-rpath,$PWD/modarith_amd -o /tmp/consumer && /tmp/consumer [chain steps]
 */
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------- "paste field.c here" ------------------------------------------- */
#ifdef USE_X448
#include "field_X448.h"
#else
#include "field_X25519.h"
#endif
/* ---------------------------------------------------------------------------------------------------------------- */

#if defined(X25519)
#define A24 121665
#define LOW_BITS_CLEARED 3
#define BASE_U 9
#elif defined(X448)
#define A24 39081
#define LOW_BITS_CLEARED 2
#define BASE_U 5
#else
#error "the pasted field must select X25519 or X448"
#endif
#if !defined(MULBYINT)
#error "the ladder multiplies by a24 with modmli"
#endif

typedef struct { spint x[Nlimbs], z[Nlimbs]; } xz;

/* RFC 7748 byte strings are little-endian, modimp / modexp take big-endian ones */
static void flip(const unsigned char *in, char *out) {
    for (int i = 0; i < Nbytes; i++) out[i] = (char)in[Nbytes - 1 - i];
}

static void step(const spint *x1, xz *p2, xz *p3) {
    spint A[Nlimbs], B[Nlimbs], C[Nlimbs], D[Nlimbs], E[Nlimbs];
    modadd(p2->x, p2->z, A);
    modsub(p2->x, p2->z, B);
    modadd(p3->x, p3->z, C);
    modsub(p3->x, p3->z, D);
    modmul(D, A, D);               /* DA */
    modmul(C, B, C);               /* CB */
    modsqr(A, A);                  /* AA, in place */
    modsqr(B, B);                  /* BB */
    modadd(D, C, p3->x);
    modsqr(p3->x, p3->x);          /* x3 = (DA + CB)^2 */
    modsub(D, C, p3->z);
    modsqr(p3->z, p3->z);
    modmul(p3->z, x1, p3->z);      /* z3 = x1 (DA - CB)^2 */
    modmul(A, B, p2->x);           /* x2 = AA BB */
    modsub(A, B, E);
    modmli(E, A24, p2->z);
    modadd(p2->z, A, p2->z);
    modmul(p2->z, E, p2->z);       /* z2 = E (AA + a24 E): output aliases an input */
}

/* out = X(k, u): RFC 7748 section 5 */
static void x_function(const unsigned char *k_in, const unsigned char *u_in, unsigned char *out) {
    unsigned char k[Nbytes], u[Nbytes];
    char be[Nbytes];
    spint x1[Nlimbs], h[Nlimbs];
    xz p2, p3;
    memcpy(k, k_in, Nbytes);
    memcpy(u, u_in, Nbytes);
    k[0] &= (unsigned char)~((1u << LOW_BITS_CLEARED) - 1);
    if (Nbits % 8) {
        k[Nbytes - 1] &= (unsigned char)((1u << (Nbits % 8)) - 1);
        u[Nbytes - 1] &= (unsigned char)((1u << (Nbits % 8)) - 1);
    }
    k[(Nbits - 1) / 8] |= (unsigned char)(1u << ((Nbits - 1) % 8));
    flip(u, be);
    modimp(be, x1);                /* values >= p are reduced by the field (RFC: "non-canonical values are accepted") */
    modone(p2.x);
    modzer(p2.z);
    modcpy(x1, p3.x);
    modone(p3.z);
    int swap = 0;
    for (int t = Nbits - 1; t >= 0; t--) {
        int kt = (k[t / 8] >> (t % 8)) & 1;
        swap ^= kt;
        modcsw(swap, p2.x, p3.x);
        modcsw(swap, p2.z, p3.z);
        swap = kt;
        step(x1, &p2, &p3);
    }
    modcsw(swap, p2.x, p3.x);
    modcsw(swap, p2.z, p3.z);
    modpro(p2.z, h);               /* the progenitor, then the inversion that uses it, in place */
    modinv(p2.z, h, p2.z);
    modmul(p2.x, p2.z, p2.x);
    modexp(p2.x, be);
    for (int i = 0; i < Nbytes; i++) out[i] = (unsigned char)be[Nbytes - 1 - i];
}

static void hex(const char *label, const unsigned char *b) {
    printf("%s ", label);
    for (int i = 0; i < Nbytes; i++) printf("%02x", b[i]);
    printf("\n");
}

static void unhex(const char *s, unsigned char *b) {
    for (int i = 0; i < Nbytes; i++) {
        unsigned v;
        sscanf(s + 2 * i, "%2x", &v);
        b[i] = (unsigned char)v;
    }
}

int main(int argc, char **argv) {
    int steps = argc > 1 ? atoi(argv[1]) : 1;
    unsigned char k[Nbytes], u[Nbytes] = {0}, v[Nbytes];
    printf("field Wordlength %d Nlimbs %d Radix %d Nbits %d Nbytes %d sizeof(spint) %d\n", Wordlength, Nlimbs, Radix, Nbits, Nbytes, (int)sizeof(spint));
    /* the RFC's own test vector (section 6.1 / 6.2): Alice's public key from her private key */
#if defined(X25519)
    unhex("77076d0a7318a57d3c16c17251b26645df4c2f87ebc0992ab177fba51db92c2a", k);
#else
    unhex("9a8f4925d1519f5775cf46b04b5800d4ee9ee8bae8bc5565d498c28dd9c9baf574a9419744897391006382a6f127ab1d9ac2d8c0a598726b", k);
#endif
    u[0] = BASE_U;
    x_function(k, u, v);
    hex("vector", v);
    /* a chain of dependent calls, keyed by a 16-bit linear congruential sequence: (k, u) -> v, (k, v) -> u, `steps` times */
    unsigned short r = 1;
    for (int i = 0; i < Nbytes; i++) {
        r = (unsigned short)(5 * r + 1);
        k[i] = (unsigned char)(r & 0xff);
    }
    memset(u, 0, Nbytes);
    u[0] = BASE_U;
    for (int i = 0; i < steps; i++) {
        x_function(k, u, v);
        x_function(k, v, u);
    }
    hex("key", k);
    printf("steps %d\n", steps);
    hex("chain", u);
    return 0;
}
