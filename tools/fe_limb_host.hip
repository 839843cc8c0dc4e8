// tools/fe_limb_host.hip -- the unsigned 32-bit-limb fields of the fused kernels (csrc/fe26.h, csrc/fe28.h) and the Edwards
// formulas built on them (csrc/ed26.h, csrc/ed28.h), compiled for the HOST into a small shared library so that
// tests/test_host_arith.py can drive them with limbs AT THE EDGES of their documented ranges (0, 1, max - 1, max) and compare with
// Python integers.  Random points never produce such limbs; the order-4 point's doubling does (round 4, Fe28::sub).
// Test tooling, not product code.
//   hipcc -O2 -std=c++17 -shared -fPIC --offload-host-only tools/fe_limb_host.hip -o /tmp/libfe_limb_host.so
#define MA_DEV __host__ __device__ inline
#include "../modarith_amd/csrc/fe26.h"
#include "../modarith_amd/csrc/fe28.h"
#include "../modarith_amd/csrc/generated/curve_ED25519.h"
#include "../modarith_amd/csrc/generated/params_X448.h"
#include "../modarith_amd/csrc/ed26.h"
#include "../modarith_amd/csrc/ed28.h"
#include "../modarith_amd/csrc/ed26s.h"
#include "../modarith_amd/csrc/ed28s.h"

using E26 = ma::Ed26<ma::C_ED25519>;
using F26 = ma::Fe26;
using E28 = ma::Ed28;
using F28 = ma::Fe28;

// field primitives: op = 0 add, 1 sub, 2 mul, 3 sqr, 4 mul_small<a24>, 5 mul_small_add<a24> (f * C + g), 6 wc (weak carry of f),
// 7 mul_k, 8 sqr_k (fe28 only); r: the result limbs as the function leaves them
extern "C" void fe26_op(int op, const uint32_t* f, const uint32_t* g, uint32_t* r) {
    switch (op) {
        case 0: F26::add(f, g, r); break;
        case 1: F26::sub(f, g, r); break;
        case 2: F26::mul(f, g, r); break;
        case 3: F26::sqr(f, r); break;
        case 4: F26::mul_small<121665>(f, r); break;
        case 5: F26::mul_small_add<121665>(f, g, r); break;
        case 6: for (int i = 0; i < 10; i++) r[i] = f[i]; E26::wc(r); break;
        default: for (int i = 0; i < 10; i++) r[i] = 0;
    }
}
extern "C" void fe28_op(int op, const uint32_t* f, const uint32_t* g, uint32_t* r) {
    switch (op) {
        case 0: F28::add(f, g, r); break;
        case 1: F28::sub(f, g, r); break;
        case 2: F28::mul(f, g, r); break;
        case 3: F28::sqr(f, r); break;
        case 4: F28::mul_small<39081>(f, r); break;
        case 5: F28::mul_small_add<39081>(f, g, r); break;
        case 6: for (int i = 0; i < 16; i++) r[i] = f[i]; E28::wc(r); break;
        case 7: F28::mul_k(f, g, r); break;
        case 8: F28::sqr_k(f, r); break;
        default: for (int i = 0; i < 16; i++) r[i] = 0;
    }
}
// canonical value as little-endian 64-bit words (4 / 7)
extern "C" void fe26_words(const uint32_t* f, uint64_t* w) { F26::to_words(f, w); }
extern "C" void fe28_words(const uint32_t* f, uint64_t* w) { F28::to_words(f, w); }
extern "C" void fe26_from_words(const uint64_t* w, uint32_t* f) { F26::from_words(w, f); }
extern "C" void fe28_from_words(const uint64_t* w, uint32_t* f) { F28::from_words(w, f); }

// Edwards formulas.  p: X, Y, Z, T (4 x NL limbs, in place).  what = 0: dbl with T, 1: add_cached(a, b, c) with T (the mixed addition of the
// fixed-base parts), 2 / 3: the Straus forms' addition of a PROJECTIVE cached entry a[0..4NL) with T, sign + / - (ed26s.h: (Y+X, Y-X, 2dT, 2Z);
// ed28s.h: (X, Y, 39081 T, Z), fetched coordinate by coordinate)
extern "C" void ed26_formula(int what, uint32_t* p, const uint32_t* a, const uint32_t* b, const uint32_t* c) {
    E26::Ext P;
    for (int i = 0; i < 10; i++) { P.X[i] = p[i]; P.Y[i] = p[10 + i]; P.Z[i] = p[20 + i]; P.T[i] = p[30 + i]; }
    if (what == 0) E26::dbl<true>(P);
    else if (what == 1) E26::add_cached<true>(P, a, b, c);
    else {
        using S = ma::Ed26Straus<ma::C_ED25519>;
        S::Cached q;
        for (int i = 0; i < 10; i++) { q.yp[i] = a[i]; q.ym[i] = a[10 + i]; q.t2d[i] = a[20 + i]; q.z2[i] = a[30 + i]; }
        if (what == 3) {        // the negated entry, as ed26s.h unpack() forms it: sums swapped, 2dT -> 2p - 2dT
            uint32_t zero[10], nt[10];
            F26::set(0, zero);
            F26::sub(zero, q.t2d, nt);
            for (int i = 0; i < 10; i++) { const uint32_t t = q.yp[i]; q.yp[i] = q.ym[i]; q.ym[i] = t; q.t2d[i] = nt[i]; }
        }
        S::add_pc(P, q, true);
    }
    for (int i = 0; i < 10; i++) { p[i] = P.X[i]; p[10 + i] = P.Y[i]; p[20 + i] = P.Z[i]; p[30 + i] = P.T[i]; }
}
extern "C" void ed28_formula(int what, uint32_t* p, const uint32_t* a, const uint32_t* b, const uint32_t* c) {
    E28::Ext P;
    for (int i = 0; i < 16; i++) { P.X[i] = p[i]; P.Y[i] = p[16 + i]; P.Z[i] = p[32 + i]; P.T[i] = p[48 + i]; }
    if (what == 0) E28::dbl<true>(P);
    else if (what == 1) E28::add_cached(P, a, b, c, true);
    else ma::Ed28Straus::add_pc(P, [&](int coord, uint32_t* out) { for (int i = 0; i < 16; i++) out[i] = a[16 * coord + i]; }, what == 3, true);
    for (int i = 0; i < 16; i++) { p[i] = P.X[i]; p[16 + i] = P.Y[i]; p[32 + i] = P.Z[i]; p[48 + i] = P.T[i]; }
}
