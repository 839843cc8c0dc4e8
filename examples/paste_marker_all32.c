/* examples/paste_marker_all32.c -- every one of the 32 functions of a generated field.c (pseudo.py:1413-1445), called through the
 * UNDECORATED names and macros that "paste field.c here" provides, on deterministic inputs; prints every result.
 *
 * tests/test_gpu_paste_marker.py compiles this file twice for each of the three BASELINE fields -- once against
 * include/field_<PRIME>.h + libmodarith_amd.so (every call one element through the GPU), once against a header that maps the same
 * names onto the CPU oracle -- and requires the two outputs to be equal line for line.  The file knows neither library: only
 * spint, Nlimbs, Nbytes, Radix and the 32 names.  Limbs of modpro / modinv / modsqrt depend on the exponentiation chain (an external
 * tool in the reference, SURVEY 8c caveat 1) and are printed after redc; everything else is printed limb for limb.
 *
 *   gcc -O2 -DFIELD_HEADER='"field_NIST256.h"' examples/paste_marker_all32.c -Iinclude -Lmodarith_amd -l:libmodarith_amd.so \
 *       -Wl,-rpath,$PWD/modarith_amd -o /tmp/all32 && /tmp/all32
 */
#include <string.h>

/* ------------------------------------------- "paste field.c here" ------------------------------------------- */
#include FIELD_HEADER
/* ---------------------------------------------------------------------------------------------------------------- */

static void show(const char *what, const spint *a) {
    printf("%-10s", what);
    for (int i = 0; i < Nlimbs; i++) printf(" %016llx", (unsigned long long)a[i]);
    printf("\n");
}
static void shown(const char *what, long long v) { printf("%-10s %lld\n", what, v); }
static void show_plain(const char *what, const spint *a) {          /* the value, canonical: chain-independent */
    spint t[Nlimbs];
    redc(a, t);
    show(what, t);
}

int main(void) {
    char ba[Nbytes], bb[Nbytes], out[Nbytes];
    spint a[Nlimbs], b[Nlimbs], c[Nlimbs], d[Nlimbs], h[Nlimbs], t[Nlimbs];
    unsigned s = 12345;
    for (int i = 0; i < Nbytes; i++) {
        s = s * 1103515245u + 12345u; ba[i] = (char)(s >> 16);
        s = s * 1103515245u + 12345u; bb[i] = (char)(s >> 16);
    }
    ba[0] &= 0x3f; bb[0] &= 0x3f;                        /* below p for every field here: modimp says 1 */
    shown("modimp", modimp(ba, a) + 2 * modimp(bb, b));
    show("a", a); show("b", b);
    modadd(a, b, c); show("modadd", c);
    modsub(a, b, c); show("modsub", c);
    modneg(b, c); show("modneg", c);
    modmul(a, b, c); show("modmul", c);
    modmul(c, b, c); show("modmul.al", c);                /* output aliases an input */
    modsqr(a, c); show("modsqr", c);
#ifdef MULBYINT
    modmli(a, 121665, c); show("modmli", c);
#endif
    modcpy(a, c); modnsqr(c, 3); show("modnsqr", c);
    modpro(a, h); show_plain("modpro", h);
    modinv(a, h, c); show_plain("modinv.h", c);
    modinv(a, NULL, d); show_plain("modinv", d);
    modmul(c, a, d); shown("inv*a==1", modis1(d));
    redc(a, c); show("redc", c);
    nres(c, d); show("nres", d);
    modzer(c); shown("modis0", modis0(c) + 2 * modis0(a));
    modone(c); shown("modis1", modis1(c) + 2 * modis1(a)); show("modone", c);
    modint(5, c); show("modint", c);
    modsqr(b, c);                                        /* a square: modqr says 1, modsqrt returns a root */
    shown("modqr", modqr(NULL, c) + 2 * modqr(NULL, a));
    modsqrt(c, NULL, d); modsqr(d, t); shown("sqrt^2==x", modcmp(t, c));
    modcpy(a, c); modcmv(0, b, c); show("modcmv0", c); modcmv(1, b, c); show("modcmv1", c);
    modcpy(a, c); modcpy(b, d); modcsw(1, c, d); show("modcsw.g", c); show("modcsw.f", d);
    redc(a, c); modshl(3, c); show("modshl", c);
    shown("modshr.r", modshr(5, c)); show("modshr", c);
    modcpy(a, c); modhaf(c); show("modhaf", c);
    mod2r(10, c); show("mod2r", c);
    modexp(a, out);
    printf("%-10s ", "modexp"); for (int i = 0; i < Nbytes; i++) printf("%02x", (unsigned char)out[i]); printf("\n");
    shown("modsign", modsign(a) + 2 * modsign(b));
    shown("modcmp", modcmp(a, a) + 2 * modcmp(a, b));
    for (int i = 0; i < Nlimbs; i++) t[i] = a[i] + b[i] + ((spint)3 << Radix);     /* limbs with excess above the radix */
    { spint m = prop(t); show("prop", t); shown("prop.mask", (long long)(m & 1)); }
    for (int i = 0; i < Nlimbs; i++) t[i] = a[i] - b[i];                            /* a negative top limb or not */
    { spint m = prop(t); shown("prop.neg", (long long)(m & 1)); spint f = flatten(t); show("flatten", t); shown("flatten.r", (long long)f); }
    modadd(a, b, t); { spint f = modfsb(t); show("modfsb", t); shown("modfsb.r", (long long)f); }
    return 0;
}
