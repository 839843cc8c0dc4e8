// tools/fe_host_check.hip -- the fused-ladder arithmetic (csrc/fe26.h, csrc/fe28.h) compiled for the HOST and run
// against the CPU oracle (oracle/liboracle.so, test infrastructure).  The device kernels wrap the very functions
// called here (x25519_fe26_one, x448_fe28_one), so a change to the limb arithmetic / ladder step can be checked in
// the build container, without a GPU, before it goes to the box.  Usage: fe_host_check [n]  (exit code 0 = all equal)
//   hipcc -O2 -std=c++17 tools/fe_host_check.hip -o /tmp/fe_host_check -Loracle -l:liboracle.so -Wl,-rpath,$PWD/oracle
#define MA_DEV __host__ __device__ inline
#include "../modarith_amd/csrc/fe26.h"
#include "../modarith_amd/csrc/fe28.h"
#include "../modarith_amd/csrc/generated/curve_ED25519.h"
#include "../modarith_amd/csrc/generated/params_NIST256.h"
#include "../modarith_amd/csrc/generated/params_X448.h"
#include "../modarith_amd/csrc/generated/params_SECP256K1.h"
#include "../modarith_amd/csrc/generated/params_NUMS256W.h"
#include "../modarith_amd/csrc/generated/params_NIST521.h"
#include "../modarith_amd/csrc/ed26.h"
#include "../modarith_amd/csrc/ed26l.h"
#include "../modarith_amd/csrc/ed26s.h"
#include "../modarith_amd/csrc/ed28.h"
#include "../modarith_amd/csrc/ed28l.h"
#include "../modarith_amd/csrc/ed28s.h"
#include "../modarith_amd/csrc/generated/curve_NIST256.h"
#include "../modarith_amd/csrc/wn26.h"
#include "../modarith_amd/csrc/generated/comb_ED25519.h"
#include "../modarith_amd/csrc/generated/comb_ED448.h"
#include "../modarith_amd/csrc/generated/curve_ED448.h"
#include "../modarith_amd/csrc/edwards.h"
#include "../modarith_amd/csrc/weierstrass.h"
#include "../modarith_amd/csrc/fh51.h"
#include "../modarith_amd/csrc/fh56.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern "C" void rfc7748_X25519(const char* bk, const char* bu, char* bv);
extern "C" void rfc7748_X448(const char* bk, const char* bu, char* bv);

static uint64_t sm_state = 0x1234567;
static uint64_t sm() {
    uint64_t z = (sm_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct pt25519 { uint64_t x[5], y[5], z[5]; };
extern "C" void ecn_ed25519_gen(pt25519*);
extern "C" void ecn_ed25519_inf(pt25519*);
extern "C" void ecn_ed25519_mul(const char* e, pt25519*);
extern "C" void ecn_ed25519_dbl(pt25519*);
extern "C" int ecn_ed25519_get(pt25519*, char* x, char* y);
extern "C" void ecn_ed25519_set(int s, const char* x, const char* y, pt25519*);

extern "C" void ecn_ed25519_mul2(const char* e, pt25519* P, const char* f, pt25519* Q, pt25519* R);

// fused double multiplication + get against the oracle's ecn mul2 + ecn get
template <class FUSED>
static int run_ed25519_mul2(const char* name, int n, FUSED fused) {
    int bad = 0;
    for (int it = 0; it < n; it++) {
        pt25519 P, Q, R;
        unsigned char e[32], f[32], k[32];
        for (int i = 0; i < 32; i++) { e[i] = (unsigned char)sm(); f[i] = (unsigned char)sm(); k[i] = (unsigned char)sm(); }
        ecn_ed25519_gen(&P); ecn_ed25519_mul((const char*)k, &P);
        for (int i = 0; i < 32; i++) k[i] = (unsigned char)sm();
        ecn_ed25519_gen(&Q); ecn_ed25519_mul((const char*)k, &Q);
        if (it % 8 == 1) ecn_ed25519_inf(&P);
        if (it % 8 == 2) ecn_ed25519_inf(&Q);
        if (it % 8 == 3) Q = P;                                    // eP + fP
        if (it % 8 == 4) { char y[32]; memset(y, 0, 32); ecn_ed25519_set(0, nullptr, y, &Q); }   // order 4
        if (it == 5) memset(e, 0, 32);
        if (it == 6) memset(f, 0, 32);
        if (it == 7) { memset(e, 0xff, 32); memset(f, 0xff, 32); }
        if (it == 13) { memset(e, 0, 32); memset(f, 0, 32); }
        if (it == 14) { memset(e, 0x88, 32); memset(f, 0x77, 32); }                    // every digit -8 (its top window carries) / -1
        if (it == 15) { memset(e, 0x80, 32); memset(f, 0x08, 32); }
        if (it == 20) { memset(e, 0, 32); e[31] = 8; memset(f, 0, 32); f[31] = 9; }
        if (it % 8 == 5 && it > 8) { char y[32]; memset(y, 0xff, 32); y[0] = 0x7f; y[31] = 0xec; ecn_ed25519_set(0, nullptr, y, &P); }   // order 2
        uint64_t ew[4], fw[4], xw[4], yw[4];
        for (int w = 0; w < 4; w++) { uint64_t v = 0, u = 0; for (int b = 0; b < 8; b++) { v |= (uint64_t)e[31 - (8 * w + b)] << (8 * b); u |= (uint64_t)f[31 - (8 * w + b)] << (8 * b); } ew[w] = v; fw[w] = u; }
        fused(ew, P.x, P.y, P.z, fw, Q.x, Q.y, Q.z, xw, yw);
        char wx[32], wy[32];
        ecn_ed25519_mul2((const char*)e, &P, (const char*)f, &Q, &R);
        ecn_ed25519_get(&R, wx, wy);
        unsigned char gx[32], gy[32];
        for (int i = 0; i < 32; i++) { gx[i] = (unsigned char)(xw[(31 - i) / 8] >> (8 * ((31 - i) % 8))); gy[i] = (unsigned char)(yw[(31 - i) / 8] >> (8 * ((31 - i) % 8))); }
        if (memcmp(gx, wx, 32) != 0 || memcmp(gy, wy, 32) != 0) { if (bad < 6) printf("%s: record %d differs\n", name, it); bad++; }
    }
    printf("%s: %d records, %d differ from the oracle's ecn mul2 + get\n", name, n, bad);
    return bad;
}

struct pt448 { uint64_t x[8], y[8], z[8]; };
extern "C" void ecn_ed448_gen(pt448*);
extern "C" void ecn_ed448_inf(pt448*);
extern "C" void ecn_ed448_mul(const char* e, pt448*);
extern "C" int ecn_ed448_get(pt448*, char* x, char* y);
extern "C" void ecn_ed448_set(int s, const char* x, const char* y, pt448*);

extern "C" void modmul_X25519(const uint64_t*, const uint64_t*, uint64_t*);
extern "C" void modsqr_X25519(const uint64_t*, uint64_t*);

// the half-limb column products of Field<P_X25519, true> (csrc/field.h pm_modmul_half / pm_modsqr_half) against the
// oracle's modmul / modsqr, limb for limb, on inputs drawn from the limb contract's edge classes (limbs < 2^53)
static int run_half(int n) {
    using F = ma::Field<ma::P_X25519, true>;
    static_assert(F::HALF, "half-limb products are expected for X25519");
    int bad = 0;
    const uint64_t edge[] = {0, 1, (1ull << 51) - 1, 1ull << 51, (1ull << 52) - 1, (1ull << 53) - 1, (1ull << 26) - 1, 1ull << 26, (1ull << 51) - 19};
    for (int it = 0; it < n; it++) {
        uint64_t a[5], b[5], got[5], want[5];
        for (int i = 0; i < 5; i++) {
            uint64_t r = sm();
            a[i] = (r % 10 < 9 && it % 3) ? edge[r % 9] : (sm() & ((1ull << 53) - 1));
            r = sm();
            b[i] = (r % 10 < 9 && it % 3 == 1) ? edge[r % 9] : (sm() & ((1ull << 53) - 1));
        }
        if (it == 0) for (int i = 0; i < 5; i++) a[i] = b[i] = (1ull << 53) - 1;
        F::modmul(a, b, got); modmul_X25519(a, b, want);
        int d = memcmp(got, want, sizeof got) != 0;
        F::modsqr(a, got); modsqr_X25519(a, want);
        d |= memcmp(got, want, sizeof got) != 0;
        F::modmul(a, b, a); modmul_X25519(want, want, want);          // aliasing compiles and runs
        if (d) { if (bad < 4) printf("half-limb modmul/modsqr: record %d differs\n", it); bad++; }
    }
    printf("Field<P_X25519,true> half-limb modmul/modsqr: %d records, %d differ from the oracle\n", n, bad);
    return bad;
}

extern "C" void modmul_NIST256(const uint64_t*, const uint64_t*, uint64_t*);
extern "C" void modsqr_NIST256(const uint64_t*, uint64_t*);
extern "C" void nres_NIST256(const uint64_t*, uint64_t*);
extern "C" void redc_NIST256(const uint64_t*, uint64_t*);

// Field<P_NIST256, true>'s Montgomery half-limb products (csrc/field.h monty_mul_half) against the oracle, limb for limb
static int run_mhalf(int n) {
    using F = ma::Field<ma::P_NIST256, true>;
    static_assert(F::MHALF, "half-limb Montgomery products are expected for NIST256");
    int bad = 0;
    const uint64_t edge[] = {0, 1, (1ull << 52) - 1, 1ull << 52, (1ull << 53) - 1, (1ull << 54) - 1, (1ull << 26) - 1, 1ull << 26, 0xffffffff0000ull};
    for (int it = 0; it < n; it++) {
        uint64_t a[5], b[5], got[5], want[5];
        for (int i = 0; i < 5; i++) {
            uint64_t r = sm();
            a[i] = (r % 10 < 9 && it % 3) ? edge[r % 9] : (sm() & ((1ull << 54) - 1));
            r = sm();
            b[i] = (r % 10 < 9 && it % 3 == 1) ? edge[r % 9] : (sm() & ((1ull << 54) - 1));
        }
        if (it == 0) for (int i = 0; i < 5; i++) a[i] = b[i] = (1ull << 54) - 1;
        int d = 0;
        F::modmul(a, b, got); modmul_NIST256(a, b, want); d |= memcmp(got, want, sizeof got) != 0;
        F::modsqr(a, got); modsqr_NIST256(a, want); d |= memcmp(got, want, sizeof got) != 0;
        F::nres(a, got); nres_NIST256(a, want); d |= memcmp(got, want, sizeof got) != 0;
        F::redc(a, got); redc_NIST256(a, want); d |= memcmp(got, want, sizeof got) != 0;
        if (d) { if (bad < 4) printf("NIST256 half-limb: record %d differs\n", it); bad++; }
    }
    printf("Field<P_NIST256,true> half-limb modmul/modsqr/nres/redc: %d records, %d differ from the oracle\n", n, bad);
    return bad;
}

extern "C" void modmul_X448(const uint64_t*, const uint64_t*, uint64_t*);
extern "C" void modsqr_X448(const uint64_t*, uint64_t*);
extern "C" void nres_X448(const uint64_t*, uint64_t*);
extern "C" void redc_X448(const uint64_t*, uint64_t*);

// Field<P_X448, true>'s trinomial half-limb products (csrc/field.h monty_mul_half_tri) against the oracle, limb for limb
static int run_mhalf448(int n) {
    using F = ma::Field<ma::P_X448, true>;
    static_assert(F::MHALF_TRI, "half-limb trinomial products are expected for X448");
    int bad = 0;
    const uint64_t edge[] = {0, 1, (1ull << 56) - 1, 1ull << 56, (1ull << 57) - 1, (1ull << 58) - 1, (1ull << 28) - 1, 1ull << 28, (1ull << 56) - 2};
    for (int it = 0; it < n; it++) {
        uint64_t a[8], b[8], got[8], want[8];
        for (int i = 0; i < 8; i++) {
            uint64_t r = sm();
            a[i] = (r % 10 < 9 && it % 3) ? edge[r % 9] : (sm() & ((1ull << 58) - 1));
            r = sm();
            b[i] = (r % 10 < 9 && it % 3 == 1) ? edge[r % 9] : (sm() & ((1ull << 58) - 1));
        }
        if (it == 0) for (int i = 0; i < 8; i++) a[i] = b[i] = (1ull << 58) - 1;
        if (it == 1) for (int i = 0; i < 8; i++) a[i] = b[i] = 0;
        int d = 0;
        F::modmul(a, b, got); modmul_X448(a, b, want); d |= memcmp(got, want, sizeof got) != 0;
        F::modsqr(a, got); modsqr_X448(a, want); d |= memcmp(got, want, sizeof got) != 0;
        F::nres(a, got); nres_X448(a, want); d |= memcmp(got, want, sizeof got) != 0;
        F::redc(a, got); redc_X448(a, want); d |= memcmp(got, want, sizeof got) != 0;
        if (d) { if (bad < 4) printf("X448 half-limb: record %d differs\n", it); bad++; }
    }
    printf("Field<P_X448,true> half-limb modmul/modsqr/nres/redc: %d records, %d differ from the oracle\n", n, bad);
    return bad;
}

// Field<P_SECP256K1, true>'s half-limb products of the "overflow" pseudo-Mersenne form against the exact 128-bit restatement
// compiled from the same header (Field<P_SECP256K1, false>, which the GPU suite pins to the reference's golden vectors)
static int run_half_ov(int n) {
    using F = ma::Field<ma::P_SECP256K1, true>;
    using X = ma::Field<ma::P_SECP256K1, false>;
    static_assert(F::HALF_OV, "half-limb overflow-form products are expected for SECP256K1");
    int bad = 0;
    const uint64_t edge[] = {0, 1, (1ull << 52) - 1, 1ull << 52, (1ull << 53) - 1, (1ull << 54) - 1, (1ull << 26) - 1, 1ull << 26, 0xffffefffffc2full};
    for (int it = 0; it < n; it++) {
        uint64_t a[5], b[5], got[5], want[5];
        for (int i = 0; i < 5; i++) {
            uint64_t r = sm();
            a[i] = (r % 10 < 9 && it % 3) ? edge[r % 9] : (sm() & ((1ull << 54) - 1));
            r = sm();
            b[i] = (r % 10 < 9 && it % 3 == 1) ? edge[r % 9] : (sm() & ((1ull << 54) - 1));
        }
        if (it == 0) for (int i = 0; i < 5; i++) a[i] = b[i] = (1ull << 54) - 1;
        int d = 0;
        F::modmul(a, b, got); X::modmul(a, b, want); d |= memcmp(got, want, sizeof got) != 0;
        F::modsqr(a, got); X::modsqr(a, want); d |= memcmp(got, want, sizeof got) != 0;
        if (d) { if (bad < 4) printf("SECP256K1 half-limb: record %d differs\n", it); bad++; }
    }
    printf("Field<P_SECP256K1,true> half-limb modmul/modsqr: %d records, %d differ from the exact products\n", n, bad);
    return bad;
}

// Field<P_NIST521, true>: the four-accumulator split products (csrc/field.h pm_modmul_split4) against the exact 128-bit rows of the
// same header (which tests/test_gpu_parity.py pins to the oracle; its NIST521 functions are bound to their parameter block at run
// time by tests/oracle_binding.py, which a C main cannot do), limbs up to the contract's edge 2^60 - 1
static int run_split4(int n) {
    using F = ma::Field<ma::P_NIST521, true>;
    using X = ma::Field<ma::P_NIST521, false>;
    static_assert(F::SPLIT4 && !X::SPLIT4, "the four-accumulator split is expected for NIST521");
    int bad = 0;
    const uint64_t Q = 1ull << 58;
    const uint64_t edge[] = {0, 1, Q - 1, Q, 2 * Q - 1, 4 * Q - 1, (1ull << 30) - 1, 1ull << 30, 4 * Q - (1ull << 30)};
    for (int it = 0; it < n; it++) {
        uint64_t a[9], b[9], got[9], want[9];
        for (int i = 0; i < 9; i++) {
            uint64_t r = sm();
            a[i] = (r % 10 < 9 && it % 3) ? edge[r % 9] : (sm() & (4 * Q - 1));
            r = sm();
            b[i] = (r % 10 < 9 && it % 3 == 1) ? edge[r % 9] : (sm() & (4 * Q - 1));
        }
        if (it == 0) for (int i = 0; i < 9; i++) a[i] = b[i] = 4 * Q - 1;
        int d = 0;
        F::modmul(a, b, got); X::modmul(a, b, want); d |= memcmp(got, want, sizeof got) != 0;
        F::modsqr(a, got); X::modsqr(a, want); d |= memcmp(got, want, sizeof got) != 0;
        F::modmul(a, b, a); d |= memcmp(a, want, 0) != 0;               // (aliasing compiles and runs)
        if (d) { if (bad < 4) printf("NIST521 split4: record %d differs\n", it); bad++; }
    }
    printf("Field<P_NIST521,true> four-accumulator split modmul/modsqr: %d records, %d differ from the exact products\n", n, bad);
    return bad;
}

// Field<P_NUMS256W, true> (half-limb columns with digit folding, mm = 0xbd0) against the exact products of the same header;
// limbs below 2^52.3: beyond 2^64 / mm the reference's own mm * a wraps and there is nothing to agree with
static int run_fold52(int n) {
    using F = ma::Field<ma::P_NUMS256W, true>;
    using X = ma::Field<ma::P_NUMS256W, false>;
    static_assert(F::FOLD52 && F::HALF_OV, "digit-folding half-limb products are expected for NUMS256W");
    int bad = 0;
    const uint64_t top = 5ull << 50;      // 2^52.32
    const uint64_t edge[] = {0, 1, (1ull << 52) - 1, 1ull << 52, top - 1, (1ull << 52) + 12345, (1ull << 26) - 1, 1ull << 26, (1ull << 52) - 189};
    for (int it = 0; it < n; it++) {
        uint64_t a[5], b[5], got[5], want[5];
        for (int i = 0; i < 5; i++) {
            uint64_t r = sm();
            a[i] = (r % 10 < 9 && it % 3) ? edge[r % 9] : (sm() % top);
            r = sm();
            b[i] = (r % 10 < 9 && it % 3 == 1) ? edge[r % 9] : (sm() % top);
        }
        if (it == 0) for (int i = 0; i < 5; i++) a[i] = b[i] = top - 1;
        int d = 0;
        F::modmul(a, b, got); X::modmul(a, b, want); d |= memcmp(got, want, sizeof got) != 0;
        F::modsqr(a, got); X::modsqr(a, want); d |= memcmp(got, want, sizeof got) != 0;
        if (d) { if (bad < 4) printf("NUMS256W half-limb: record %d differs\n", it); bad++; }
    }
    printf("Field<P_NUMS256W,true> half-limb modmul/modsqr: %d records, %d differ from the exact products\n", n, bad);
    return bad;
}

extern "C" void ecn_ed448_mul2(const char* e, pt448* P, const char* f, pt448* Q, pt448* R);

static int run_ed448_mul2(int n) {
    int bad = 0;
    for (int it = 0; it < n; it++) {
        pt448 P, Q, R;
        unsigned char e[56], f[56], k[56];
        for (int i = 0; i < 56; i++) { e[i] = (unsigned char)sm(); f[i] = (unsigned char)sm(); k[i] = (unsigned char)sm(); }
        ecn_ed448_gen(&P); ecn_ed448_mul((const char*)k, &P);
        for (int i = 0; i < 56; i++) k[i] = (unsigned char)sm();
        ecn_ed448_gen(&Q); ecn_ed448_mul((const char*)k, &Q);
        if (it % 8 == 1) ecn_ed448_inf(&P);
        if (it % 8 == 2) ecn_ed448_inf(&Q);
        if (it % 8 == 3) Q = P;
        if (it % 8 == 4) { char y[56]; memset(y, 0, 56); ecn_ed448_set(0, nullptr, y, &Q); }
        if (it == 5) memset(e, 0, 56);
        if (it == 6) memset(f, 0, 56);
        if (it == 7) { memset(e, 0xff, 56); memset(f, 0xff, 56); }
        if (it == 14) { memset(e, 0x88, 56); memset(f, 0x77, 56); }                    // every digit -8 (the top window carries) / -1
        if (it == 15) { memset(e, 0x80, 56); memset(f, 0x08, 56); }
        if (it == 13) { memset(e, 0, 56); memset(f, 0, 56); }
        if (it % 8 == 5 && it > 8) { char y[56]; memset(y, 0xff, 56); y[27] = (char)0xfe; y[55] = (char)0xfe; ecn_ed448_set(0, nullptr, y, &P); }   // order 2
        uint64_t ew[7], fw[7], xw[7], yw[7];
        for (int w = 0; w < 7; w++) { uint64_t v = 0, u = 0; for (int b = 0; b < 8; b++) { v |= (uint64_t)e[55 - (8 * w + b)] << (8 * b); u |= (uint64_t)f[55 - (8 * w + b)] << (8 * b); } ew[w] = v; fw[w] = u; }
        ma::ed448_mul2_get_straus_one(ew, P.x, P.y, P.z, fw, Q.x, Q.y, Q.z, xw, yw);
        char wx[56], wy[56];
        ecn_ed448_mul2((const char*)e, &P, (const char*)f, &Q, &R);
        ecn_ed448_get(&R, wx, wy);
        unsigned char gx[56], gy[56];
        for (int i = 0; i < 56; i++) { gx[i] = (unsigned char)(xw[(55 - i) / 8] >> (8 * ((55 - i) % 8))); gy[i] = (unsigned char)(yw[(55 - i) / 8] >> (8 * ((55 - i) % 8))); }
        if (memcmp(gx, wx, 56) != 0 || memcmp(gy, wy, 56) != 0) { if (bad < 6) printf("ed448_mul2_get_straus_one: record %d differs\n", it); bad++; }
    }
    printf("ed448_mul2_get_straus_one: %d records, %d differ from the oracle's ecn mul2 + get\n", n, bad);
    return bad;
}

template <int NW, class Fn, class Ref>
static int run(const char* name, int n, Fn fn, Ref ref, uint64_t fixed_u = 0) {
    int bad = 0;
    for (int it = 0; it < n; it++) {
        uint64_t k[NW], u[NW], got[NW], want[NW];
        for (int i = 0; i < NW; i++) { k[i] = sm(); u[i] = sm(); }
        // corner records first: u = 0, 1, all ones, p-1 .. p+1 region (top words all ones), scalar all zeros / all ones
        if (it == 0) memset(u, 0, sizeof u);
        if (it == 1) { memset(u, 0, sizeof u); u[0] = 1; }
        if (it == 2) memset(u, 0xff, sizeof u);
        if (it == 3) { memset(u, 0xff, sizeof u); u[0] = ~(uint64_t)18; if (NW == 4) u[3] >>= 1; }
        if (it == 4) { memset(u, 0xff, sizeof u); u[0] = ~(uint64_t)19; if (NW == 4) u[3] >>= 1; }
        if (it == 5) memset(k, 0, sizeof k);
        if (it == 6) memset(k, 0xff, sizeof k);
        if (it == 7) { memset(u, 0, sizeof u); u[0] = (NW == 4) ? 9 : 5; }
        if (fixed_u) {                                       // base-point runs: u is the base point, more scalar corners
            memset(u, 0, sizeof u);
            u[0] = fixed_u;
            if (it == 8 && NW == 7) {                        // 4q: the clamped scalar that reaches the neutral element of ED448 -> 0
                const uint64_t q4[7] = {0x8de30a4aad6113ccull, 0x85b309ca37163d54ull, 0x113b6d26bb58da40ull, 0xfffffffdf3288fa7ull, ~0ull, ~0ull, ~0ull};
                memcpy(k, q4, sizeof q4);
            }
            if (it == 9) { memset(k, 0x88, sizeof k); }
            if (it == 10) { memset(k, 0x77, sizeof k); }
        }
        fn(k, u, got);
        ref((const char*)k, (const char*)u, (char*)want);
        if (memcmp(got, want, sizeof got) != 0) {
            if (bad < 4) printf("%s: record %d differs\n", name, it);
            bad++;
        }
    }
    printf("%s: %d records, %d differ from the oracle\n", name, n, bad);
    return bad;
}

struct pt256 { uint64_t x[5], y[5], z[5]; };
extern "C" void ecn_nist256_gen(pt256*);
extern "C" void ecn_nist256_inf(pt256*);
extern "C" void ecn_nist256_neg(pt256*);
extern "C" void ecn_nist256_mul(const char* e, pt256*);
extern "C" void ecn_nist256_mul2(const char* e, pt256* P, const char* f, pt256* Q, pt256* R);
extern "C" int ecn_nist256_get(pt256*, char* x, char* y);

static void be_words4(const unsigned char* e, uint64_t* w) {
    for (int k = 0; k < 4; k++) { uint64_t v = 0; for (int b = 0; b < 8; b++) v |= (uint64_t)e[31 - (8 * k + b)] << (8 * b); w[k] = v; }
}
static void words4_be(const uint64_t* w, unsigned char* o) {
    for (int i = 0; i < 32; i++) o[i] = (unsigned char)(w[(31 - i) / 8] >> (8 * ((31 - i) % 8)));
}
static const unsigned char NIST256_ORDER[32] = {0xff,0xff,0xff,0xff,0,0,0,0,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xbc,0xe6,0xfa,0xad,0xa7,0x17,0x9e,0x84,0xf3,0xb9,0xca,0xc2,0xfc,0x63,0x25,0x51};

// fused P-256 mul+get (csrc/wn26.h) against the oracle's ecn mul followed by ecn get: random projective points (random
// multiples of the generator, NOT normalised), the point at infinity, the affine generator; scalars 0, 1, 2, the group
// order q, q - 1, q + 1, all ones, single-window values
static int run_nist256(int n) {
    int bad = 0;
    for (int it = 0; it < n; it++) {
        pt256 P;
        unsigned char e[32], k[32];
        for (int i = 0; i < 32; i++) { e[i] = (unsigned char)sm(); k[i] = (unsigned char)sm(); }
        ecn_nist256_gen(&P);
        ecn_nist256_mul((const char*)k, &P);
        if (it % 16 == 1) ecn_nist256_inf(&P);
        if (it % 16 == 4) ecn_nist256_gen(&P);
        if (it == 5 || it == 17) memset(e, 0, 32);
        if (it == 6) { memset(e, 0, 32); e[31] = 1; }
        if (it == 7) memset(e, 0xff, 32);
        if (it == 8) memcpy(e, NIST256_ORDER, 32);
        if (it == 9) { memcpy(e, NIST256_ORDER, 32); e[31] -= 1; }
        if (it == 10) { memcpy(e, NIST256_ORDER, 32); e[31] += 1; }
        if (it == 11) { memset(e, 0, 32); e[31] = 2; }
        if (it == 12) { memset(e, 0, 32); e[31] = 8; }
        if (it == 13) { memset(e, 0, 32); e[31] = 0x88; }
        if (it == 14) { memset(e, 0, 32); e[0] = 0x80; }
        pt256 Q = P;
        uint64_t ew[4], xw[4], yw[4], tab[ma::NIST256_TABLE_WORDS];
        be_words4(e, ew);
        ma::nist256_mul_get_one<ma::C_NIST256>(ew, P.x, P.y, P.z, tab, 1, xw, yw);
        char wx[32], wy[32];
        ecn_nist256_mul((const char*)e, &Q);
        ecn_nist256_get(&Q, wx, wy);
        unsigned char gx[32], gy[32];
        words4_be(xw, gx);
        words4_be(yw, gy);
        if (memcmp(gx, wx, 32) != 0 || memcmp(gy, wy, 32) != 0) {
            if (bad < 6) printf("nist256_mul_get_one: record %d differs\n", it);
            bad++;
        }
    }
    printf("nist256_mul_get_one: %d records, %d differ from the oracle's ecn mul + get\n", n, bad);
    return bad;
}

// fused P-256 mul2+get against the oracle's ecn mul2 + get; includes e*P + f*Q = infinity (Q = -P, f = e), P = Q, infinite inputs
static int run_nist256_mul2(int n) {
    int bad = 0;
    for (int it = 0; it < n; it++) {
        pt256 P, Q, R;
        unsigned char e[32], f[32], k[32];
        for (int i = 0; i < 32; i++) { e[i] = (unsigned char)sm(); f[i] = (unsigned char)sm(); k[i] = (unsigned char)sm(); }
        ecn_nist256_gen(&P); ecn_nist256_mul((const char*)k, &P);
        for (int i = 0; i < 32; i++) k[i] = (unsigned char)sm();
        ecn_nist256_gen(&Q); ecn_nist256_mul((const char*)k, &Q);
        if (it % 16 == 1) ecn_nist256_inf(&P);
        if (it % 16 == 2) ecn_nist256_inf(&Q);
        if (it % 16 == 3) { Q = P; ecn_nist256_neg(&Q); memcpy(f, e, 32); }     // e P + e (-P) = infinity
        if (it % 16 == 4) Q = P;
        if (it % 16 == 5) { Q = P; ecn_nist256_neg(&Q); }
        if (it == 6) { memset(e, 0, 32); memset(f, 0, 32); }
        if (it == 7) { memset(e, 0xff, 32); memset(f, 0xff, 32); }
        if (it == 8) { memcpy(e, NIST256_ORDER, 32); memset(f, 0, 32); f[31] = 1; }
        pt256 P0 = P, Q0 = Q;
        uint64_t ew[4], fw[4], xw[4], yw[4], tab[ma::NIST256_TABLE_WORDS];
        be_words4(e, ew);
        be_words4(f, fw);
        ma::nist256_mul2_get_one<ma::C_NIST256>(ew, P.x, P.y, P.z, fw, Q.x, Q.y, Q.z, tab, 1, xw, yw);
        char wx[32], wy[32];
        ecn_nist256_mul2((const char*)e, &P0, (const char*)f, &Q0, &R);
        ecn_nist256_get(&R, wx, wy);
        unsigned char gx[32], gy[32];
        words4_be(xw, gx);
        words4_be(yw, gy);
        if (memcmp(gx, wx, 32) != 0 || memcmp(gy, wy, 32) != 0) { if (bad < 6) printf("nist256_mul2_get_one: record %d differs\n", it); bad++; }
    }
    printf("nist256_mul2_get_one: %d records, %d differ from the oracle's ecn mul2 + get\n", n, bad);
    return bad;
}

// fused generator multiplication on the Edwards curves (fixed-base tables generated/comb_<C>.h as host arrays) against the
// oracle's ecn gen + ecn mul + ecn get: corner scalars (0, 1, single windows, 8 / 9 nibbles, all ones) and random ones
static const int32_t comb_ed25519_host[] = { COMB_ED25519_VALUES };
static const int32_t comb_ed448_host[] = { COMB_ED448_VALUES };
struct HostComb25519 { static constexpr int W = COMB_ED25519_W, NW = COMB_ED25519_WINDOWS; static int32_t get(int idx) { return comb_ed25519_host[idx]; } };
struct HostComb448 { static constexpr int W = COMB_ED448_W, NW = COMB_ED448_WINDOWS; static int32_t get(int idx) { return comb_ed448_host[idx]; } };
template <int NB, class PT, class FN, class GEN, class MUL, class GET>
static int run_edgen(const char* name, int n, FN fused, GEN gen, MUL mul, GET get) {
    int bad = 0;
    for (int it = 0; it < n; it++) {
        unsigned char e[NB];
        for (int i = 0; i < NB; i++) e[i] = (unsigned char)sm();
        if (it == 0) memset(e, 0, NB);
        if (it == 1) { memset(e, 0, NB); e[NB - 1] = 1; }
        if (it == 2) memset(e, 0xff, NB);
        if (it == 3) memset(e, 0x88, NB);
        if (it == 4) memset(e, 0x99, NB);
        if (it == 5) memset(e, 0x77, NB);
        if (it == 6) { memset(e, 0, NB); e[NB - 1] = 8; }
        if (it == 7) { memset(e, 0, NB); e[0] = 0x80; }
        if (it == 8) { memset(e, 0, NB); e[NB - 1] = 0x10; }
        if (it >= 9 && it < 9 + 2 * NB) { memset(e, 0, NB); e[(it - 9) / 2] = (it & 1) ? 0x08 : 0x90; }
        uint64_t ew[NB / 8], xw[NB / 8], yw[NB / 8];
        for (int w = 0; w < NB / 8; w++) { uint64_t v = 0; for (int b = 0; b < 8; b++) v |= (uint64_t)e[NB - 1 - (8 * w + b)] << (8 * b); ew[w] = v; }
        fused(ew, xw, yw);
        PT P;
        gen(&P);
        mul((const char*)e, &P);
        char wx[NB], wy[NB];
        get(&P, wx, wy);
        unsigned char gx[NB], gy[NB];
        for (int i = 0; i < NB; i++) { gx[i] = (unsigned char)(xw[(NB - 1 - i) / 8] >> (8 * ((NB - 1 - i) % 8))); gy[i] = (unsigned char)(yw[(NB - 1 - i) / 8] >> (8 * ((NB - 1 - i) % 8))); }
        if (memcmp(gx, wx, NB) != 0 || memcmp(gy, wy, NB) != 0) { if (bad < 6) printf("%s: record %d differs\n", name, it); bad++; }
    }
    printf("%s: %d records, %d differ from the oracle's ecn gen + mul + get\n", name, n, bad);
    return bad;
}

// ---- round 5: the ladder form of the fused ED25519 multiplications (csrc/ed26l.h) against the oracle's ecn mul (+ mul2) + get.
// The window form's complete formulas have no exceptional pair (P, e); the ladder + recovery has four classes of them
// (P in {O, T2}; [e]P in {O, T2}; [e]P = -P), so the records are built to hit every class: P = [k]G + S for every S of the
// 8-torsion subgroup (orders 1, 2, 4, 4, 8, 8, 8, 8 -- and the torsion points themselves, k = 0), e = j q - 1, j q, j q + 1
// for j = 0 .. 8 (q the prime group order: [j q]P runs through the torsion part of P), 0, 1, 2, 2^256 - 1, random.
extern "C" void ecn_ed25519_add(pt25519*, pt25519*);
extern "C" int ecn_ed25519_isinf(const pt25519*);
static const unsigned char ED25519_Q[32] = {0x10,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0x14,0xde,0xf9,0xde,0xa2,0xf7,0x9c,0xd6,0x58,0x12,0x63,0x1a,0x5c,0xf5,0xd3,0xed};
static void ed25519_jq(int j, int delta, unsigned char* e) {      // e = j q + delta (big-endian, mod 2^256), delta in {-1, 0, 1}
    unsigned __int128 acc = 0;
    uint64_t w[4], qw[4];
    for (int k = 0; k < 4; k++) { uint64_t v = 0; for (int b = 0; b < 8; b++) v |= (uint64_t)ED25519_Q[31 - (8 * k + b)] << (8 * b); qw[k] = v; }
    for (int k = 0; k < 4; k++) { acc += (unsigned __int128)qw[k] * (unsigned)j; w[k] = (uint64_t)acc; acc >>= 64; }
    if (delta > 0) { for (int k = 0; k < 4 && ++w[k] == 0; k++) {} }
    if (delta < 0) { for (int k = 0; k < 4 && w[k]-- == 0; k++) {} }
    for (int i = 0; i < 32; i++) e[31 - i] = (unsigned char)(w[i / 8] >> (8 * (i % 8)));
}
static bool ed25519_torsion8(pt25519* T8) {                       // a point of order 8: [q] of a decompressed point, until 4 T != O
    for (int y0 = 2; y0 < 200; y0++) {
        char y[32];
        memset(y, 0, 32);
        y[31] = (char)y0;
        pt25519 P;
        ecn_ed25519_set(0, nullptr, y, &P);
        if (ecn_ed25519_isinf(&P)) continue;                      // not a y-coordinate of the curve
        ecn_ed25519_mul((const char*)ED25519_Q, &P);
        pt25519 D = P;
        ecn_ed25519_dbl(&D); ecn_ed25519_dbl(&D);
        if (!ecn_ed25519_isinf(&D)) { *T8 = P; return true; }
    }
    return false;
}
template <class FUSED>
static int run_ed25519_lad(const char* name, int n, FUSED fused) {
    int bad = 0, it = 0;
    pt25519 T8;
    if (!ed25519_torsion8(&T8)) { printf("%s: no point of order 8 found\n", name); return 1; }
    pt25519 S[8];
    ecn_ed25519_inf(&S[0]);
    for (int j = 1; j < 8; j++) { S[j] = S[j - 1]; ecn_ed25519_add(&T8, &S[j]); }
    auto one = [&](const pt25519& P0, const unsigned char* e) {
        pt25519 P = P0, Q = P0;
        uint64_t ew[4], xw[4], yw[4];
        for (int w = 0; w < 4; w++) { uint64_t v = 0; for (int b = 0; b < 8; b++) v |= (uint64_t)e[31 - (8 * w + b)] << (8 * b); ew[w] = v; }
        fused(ew, P.x, P.y, P.z, xw, yw);
        char wx[32], wy[32];
        ecn_ed25519_mul((const char*)e, &Q);
        ecn_ed25519_get(&Q, wx, wy);
        unsigned char gx[32], gy[32];
        for (int i = 0; i < 32; i++) { gx[i] = (unsigned char)(xw[(31 - i) / 8] >> (8 * ((31 - i) % 8))); gy[i] = (unsigned char)(yw[(31 - i) / 8] >> (8 * ((31 - i) % 8))); }
        if (memcmp(gx, wx, 32) != 0 || memcmp(gy, wy, 32) != 0) { if (bad < 8) printf("%s: record %d differs\n", name, it); bad++; }
        it++;
    };
    for (int tors = 0; tors < 8; tors++) {
        for (int base = 0; base < 3; base++) {                    // S alone; [k]G + S projective; the same made affine (Z = 1)
            pt25519 P = S[tors];
            if (base) {
                unsigned char k[32];
                for (int i = 0; i < 32; i++) k[i] = (unsigned char)sm();
                pt25519 B;
                ecn_ed25519_gen(&B);
                ecn_ed25519_mul((const char*)k, &B);
                ecn_ed25519_add(&B, &P);
                if (base == 2) { char x[32], y[32]; ecn_ed25519_get(&P, x, y); ecn_ed25519_set(0, x, y, &P); }
            }
            unsigned char e[32];
            for (int j = 0; j <= 8; j++) for (int dl = -1; dl <= 1; dl++) { ed25519_jq(j, dl, e); one(P, e); }
            memset(e, 0, 32); e[31] = 2; one(P, e);
            memset(e, 0xff, 32); one(P, e);
            memset(e, 0, 32); e[0] = 0x80; one(P, e);
            for (int r = 0; r < 2; r++) { for (int i = 0; i < 32; i++) e[i] = (unsigned char)sm(); one(P, e); }
        }
    }
    for (int r = 0; r < n; r++) {                                 // random projective points (not normalised), random scalars
        unsigned char e[32], k[32];
        for (int i = 0; i < 32; i++) { e[i] = (unsigned char)sm(); k[i] = (unsigned char)sm(); }
        pt25519 P;
        ecn_ed25519_gen(&P);
        ecn_ed25519_mul((const char*)k, &P);
        if (r % 16 == 4) ecn_ed25519_gen(&P);
        one(P, e);
    }
    printf("%s: %d records, %d differ from the oracle's ecn mul + get\n", name, it, bad);
    return bad;
}


// e*G + f*Q in the ladder form (Ed26Lad::mulgen2_get_one) against the oracle's gen, mul2, get: Q and f through the exceptional classes of
// run_ed25519_lad (f*Q by ladder + recovery), e random / 0 / -f-ish corner values (the fixed-base part adds with complete formulas)
static int run_ed25519_mulgen2_lad(int n) {
    int bad = 0, it = 0;
    pt25519 T8;
    if (!ed25519_torsion8(&T8)) { printf("Ed26Lad::mulgen2_get_one: no point of order 8 found\n"); return 1; }
    pt25519 S[8];
    ecn_ed25519_inf(&S[0]);
    for (int j = 1; j < 8; j++) { S[j] = S[j - 1]; ecn_ed25519_add(&T8, &S[j]); }
    auto one = [&](const pt25519& Q0, const unsigned char* e, const unsigned char* f) {
        pt25519 G, Q = Q0, Qc = Q0, R;
        ecn_ed25519_gen(&G);
        uint64_t ew[4], fw[4], xw[4], yw[4];
        for (int w = 0; w < 4; w++) { uint64_t v = 0, u = 0; for (int b = 0; b < 8; b++) { v |= (uint64_t)e[31 - (8 * w + b)] << (8 * b); u |= (uint64_t)f[31 - (8 * w + b)] << (8 * b); } ew[w] = v; fw[w] = u; }
        ma::Ed26Lad<ma::C_ED25519>::mulgen2_get_one<HostComb25519>(ew, fw, Q.x, Q.y, Q.z, xw, yw);
        char wx[32], wy[32];
        ecn_ed25519_mul2((const char*)e, &G, (const char*)f, &Qc, &R);
        ecn_ed25519_get(&R, wx, wy);
        unsigned char gx[32], gy[32];
        for (int i = 0; i < 32; i++) { gx[i] = (unsigned char)(xw[(31 - i) / 8] >> (8 * ((31 - i) % 8))); gy[i] = (unsigned char)(yw[(31 - i) / 8] >> (8 * ((31 - i) % 8))); }
        if (memcmp(gx, wx, 32) != 0 || memcmp(gy, wy, 32) != 0) { if (bad < 8) printf("Ed26Lad::mulgen2_get_one: record %d differs\n", it); bad++; }
        it++;
    };
    for (int tors = 0; tors < 8; tors++) {
        for (int base = 0; base < 2; base++) {
            pt25519 Q = S[tors];
            if (base) {
                unsigned char k[32];
                for (int i = 0; i < 32; i++) k[i] = (unsigned char)sm();
                pt25519 B;
                ecn_ed25519_gen(&B);
                ecn_ed25519_mul((const char*)k, &B);
                ecn_ed25519_add(&B, &Q);
            }
            unsigned char e[32], f[32];
            for (int j = 0; j <= 8; j += (tors & 1) ? 1 : 2) for (int dl = -1; dl <= 1; dl++) {
                ed25519_jq(j, dl, f);
                for (int i = 0; i < 32; i++) e[i] = (unsigned char)sm();
                if (dl == 0) memset(e, 0, 32);
                one(Q, e, f);
            }
            memset(f, 0xff, 32); memset(e, 0xff, 32); one(Q, e, f);
        }
    }
    for (int r = 0; r < n; r++) {
        unsigned char e[32], f[32], k[32];
        for (int i = 0; i < 32; i++) { e[i] = (unsigned char)sm(); f[i] = (unsigned char)sm(); k[i] = (unsigned char)sm(); }
        pt25519 Q;
        ecn_ed25519_gen(&Q);
        if (r % 8 != 3) ecn_ed25519_mul((const char*)k, &Q);      // (r % 8 == 3: Q = G; then f = q - e makes the sum neutral)
        if (r % 8 == 3 && r % 16 == 3) {
            uint64_t ew[4], qw[4], fw[4];
            for (int w = 0; w < 4; w++) { uint64_t v = 0, u = 0; for (int b = 0; b < 8; b++) { v |= (uint64_t)e[31 - (8 * w + b)] << (8 * b); u |= (uint64_t)ED25519_Q[31 - (8 * w + b)] << (8 * b); } ew[w] = v; qw[w] = u; }
            ew[3] &= 0x0fffffffffffffffull;                        // e < q
            unsigned __int128 br = 0;
            for (int w = 0; w < 4; w++) { unsigned __int128 d = (unsigned __int128)qw[w] - ew[w] - (uint64_t)br; fw[w] = (uint64_t)d; br = (d >> 64) & 1; }
            for (int i = 0; i < 32; i++) { e[31 - i] = (unsigned char)(ew[i / 8] >> (8 * (i % 8))); f[31 - i] = (unsigned char)(fw[i / 8] >> (8 * (i % 8))); }
        }
        one(Q, e, f);
    }
    printf("Ed26Lad::mulgen2_get_one: %d records, %d differ from the oracle's ecn gen + mul2 + get\n", it, bad);
    return bad;
}


// ---- round 5: the ladder form of the fused ED448 multiplication (csrc/ed28l.h), records built like run_ed25519_lad's: P = [k]G + S for
// every S of the 4-torsion subgroup (orders 1, 4, 2, 4) and the torsion points themselves, e = j q - 1, j q, j q + 1 for j = 0 .. 4,
// 0 / 2 / all ones / top bit, random
extern "C" void ecn_ed448_add(pt448*, pt448*);
static const unsigned char ED448_Q[56] = {0x3f,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,
                                          0x7c,0xca,0x23,0xe9,0xc4,0x4e,0xdb,0x49,0xae,0xd6,0x36,0x90,0x21,0x6c,0xc2,0x72,0x8d,0xc5,0x8f,0x55,0x23,0x78,0xc2,0x92,0xab,0x58,0x44,0xf3};
static void ed448_jq(int j, int delta, unsigned char* e) {
    unsigned __int128 acc = 0;
    uint64_t w[7], qw[7];
    for (int k = 0; k < 7; k++) { uint64_t v = 0; for (int b = 0; b < 8; b++) v |= (uint64_t)ED448_Q[55 - (8 * k + b)] << (8 * b); qw[k] = v; }
    for (int k = 0; k < 7; k++) { acc += (unsigned __int128)qw[k] * (unsigned)j; w[k] = (uint64_t)acc; acc >>= 64; }
    if (delta > 0) { for (int k = 0; k < 7 && ++w[k] == 0; k++) {} }
    if (delta < 0) { for (int k = 0; k < 7 && w[k]-- == 0; k++) {} }
    for (int i = 0; i < 56; i++) e[55 - i] = (unsigned char)(w[i / 8] >> (8 * (i % 8)));
}
static int run_ed448_lad(int n) {
    int bad = 0, it = 0;
    pt448 T4, S[4];
    { char y[56]; memset(y, 0, 56); ecn_ed448_set(0, nullptr, y, &T4); }                  // y = 0: a point of order 4
    ecn_ed448_inf(&S[0]);
    for (int j = 1; j < 4; j++) { S[j] = S[j - 1]; ecn_ed448_add(&T4, &S[j]); }
    auto one = [&](const pt448& P0, const unsigned char* e) {
        pt448 P = P0, Q = P0;
        uint64_t ew[7], xw[7], yw[7];
        for (int w = 0; w < 7; w++) { uint64_t v = 0; for (int b = 0; b < 8; b++) v |= (uint64_t)e[55 - (8 * w + b)] << (8 * b); ew[w] = v; }
        ma::Ed28Lad::mul_get_one(ew, P.x, P.y, P.z, xw, yw);
        char wx[56], wy[56];
        ecn_ed448_mul((const char*)e, &Q);
        ecn_ed448_get(&Q, wx, wy);
        unsigned char gx[56], gy[56];
        for (int i = 0; i < 56; i++) { gx[i] = (unsigned char)(xw[(55 - i) / 8] >> (8 * ((55 - i) % 8))); gy[i] = (unsigned char)(yw[(55 - i) / 8] >> (8 * ((55 - i) % 8))); }
        if (memcmp(gx, wx, 56) != 0 || memcmp(gy, wy, 56) != 0) { if (bad < 8) printf("Ed28Lad::mul_get_one: record %d differs\n", it); bad++; }
        it++;
    };
    for (int tors = 0; tors < 4; tors++) {
        for (int base = 0; base < 3; base++) {
            pt448 P = S[tors];
            if (base) {
                unsigned char k[56];
                for (int i = 0; i < 56; i++) k[i] = (unsigned char)sm();
                pt448 B;
                ecn_ed448_gen(&B);
                ecn_ed448_mul((const char*)k, &B);
                ecn_ed448_add(&B, &P);
                if (base == 2) { char x[56], y[56]; ecn_ed448_get(&P, x, y); ecn_ed448_set(0, x, y, &P); }
            }
            unsigned char e[56];
            for (int j = 0; j <= 4; j++) for (int dl = -1; dl <= 1; dl++) { ed448_jq(j, dl, e); one(P, e); }
            memset(e, 0, 56); e[55] = 2; one(P, e);
            memset(e, 0xff, 56); one(P, e);
            memset(e, 0, 56); e[0] = 0x80; one(P, e);
            for (int i = 0; i < 56; i++) e[i] = (unsigned char)sm();
            one(P, e);
        }
    }
    for (int r = 0; r < n; r++) {
        unsigned char e[56], k[56];
        for (int i = 0; i < 56; i++) { e[i] = (unsigned char)sm(); k[i] = (unsigned char)sm(); }
        pt448 P;
        ecn_ed448_gen(&P);
        if (r % 8 != 4) ecn_ed448_mul((const char*)k, &P);
        one(P, e);
    }
    printf("Ed28Lad::mul_get_one: %d records, %d differ from the oracle's ecn mul + get\n", it, bad);
    return bad;
}

extern "C" void ecn_ed448_mul2(const char* e, pt448* P, const char* f, pt448* Q, pt448* R);
static int run_ed448_mulgen2(int n) {
    int bad = 0;
    for (int it = 0; it < n; it++) {
        pt448 G, Q, R;
        unsigned char e[56], f[56], k[56];
        for (int i = 0; i < 56; i++) { e[i] = (unsigned char)sm(); f[i] = (unsigned char)sm(); k[i] = (unsigned char)sm(); }
        ecn_ed448_gen(&G);
        ecn_ed448_gen(&Q); ecn_ed448_mul((const char*)k, &Q);
        if (it % 8 == 1) ecn_ed448_inf(&Q);
        if (it % 8 == 2) { char y[56]; memset(y, 0, 56); ecn_ed448_set(0, nullptr, y, &Q); }            // order 4
        if (it % 8 == 3) ecn_ed448_gen(&Q);
        if (it % 8 == 4) memset(e, 0, 56);
        if (it % 8 == 5) memset(f, 0, 56);
        if (it == 6) { memset(e, 0xff, 56); memset(f, 0xff, 56); }
        pt448 Q0 = Q;
        if (it >= 9 && it < 21) ed448_jq((it - 9) / 3, (it - 9) % 3 - 1, f);          // f = j q - 1, j q, j q + 1: f*Q runs through -Q, the neutral element, Q (+ torsion)
        uint64_t ew[7], fw[7], xw[7], yw[7];
        for (int w = 0; w < 7; w++) { uint64_t v = 0, u = 0; for (int b = 0; b < 8; b++) { v |= (uint64_t)e[55 - (8 * w + b)] << (8 * b); u |= (uint64_t)f[55 - (8 * w + b)] << (8 * b); } ew[w] = v; fw[w] = u; }
        ma::Ed28Lad::mulgen2_get_one<HostComb448>(ew, fw, Q.x, Q.y, Q.z, xw, yw);
        char wx[56], wy[56];
        ecn_ed448_mul2((const char*)e, &G, (const char*)f, &Q0, &R);
        ecn_ed448_get(&R, wx, wy);
        unsigned char gx[56], gy[56];
        for (int i = 0; i < 56; i++) { gx[i] = (unsigned char)(xw[(55 - i) / 8] >> (8 * ((55 - i) % 8))); gy[i] = (unsigned char)(yw[(55 - i) / 8] >> (8 * ((55 - i) % 8))); }
        if (memcmp(gx, wx, 56) != 0 || memcmp(gy, wy, 56) != 0) { if (bad < 6) printf("Ed28Lad::mulgen2_get_one: record %d differs\n", it); bad++; }
    }
    printf("Ed28Lad::mulgen2_get_one: %d records, %d differ from the oracle's ecn gen + mul2 + get\n", n, bad);
    return bad;
}


// ---------------------------------------------------------------------------------------------------------------------------
// Round 4: the half-limb RESIDENT form of 2^255-19 (csrc/fh51.h) and the "_u" sums (csrc/field.h modadd_u / modsub_u / modneg_u)
// against the oracle, limb for limb; then the curve formulas built on them (csrc/edwards.h on FieldH51, csrc/weierstrass.h with
// "_u" sums) against the oracle's ecn add / dbl on random projective points.
extern "C" void modadd_X25519(const uint64_t*, const uint64_t*, uint64_t*);
extern "C" void modsub_X25519(const uint64_t*, const uint64_t*, uint64_t*);
extern "C" void modneg_X25519(const uint64_t*, uint64_t*);
extern "C" void modadd_NIST256(const uint64_t*, const uint64_t*, uint64_t*);
extern "C" void modsub_NIST256(const uint64_t*, const uint64_t*, uint64_t*);
extern "C" void modneg_NIST256(const uint64_t*, uint64_t*);
extern "C" void modadd_X448(const uint64_t*, const uint64_t*, uint64_t*);
extern "C" void modsub_X448(const uint64_t*, const uint64_t*, uint64_t*);
extern "C" void modneg_X448(const uint64_t*, uint64_t*);

static int run_fh51(int n) {
    using H = ma::FieldH51<ma::P_X25519>;
    int bad = 0;
    const uint64_t edge[] = {0, 1, (1ull << 51) - 1, 1ull << 51, (1ull << 52) - 1, (1ull << 53) - 1, (1ull << 26) - 1, 1ull << 26, (1ull << 51) - 19, 37, 38, 39, (1ull << 51) - 38};
    for (int it = 0; it < n; it++) {
        uint64_t a[5], b[5], c[5], got[5], want[5], t[5];
        for (int i = 0; i < 5; i++) {
            uint64_t r = sm();
            a[i] = (r % 10 < 8 && it % 3) ? edge[r % 13] : (sm() & ((1ull << 53) - 1));
            r = sm();
            b[i] = (r % 10 < 8 && it % 3 == 1) ? edge[r % 13] : (sm() & ((1ull << 53) - 1));
            c[i] = sm() & ((1ull << 52) - 1);
        }
        if (it == 0) for (int i = 0; i < 5; i++) a[i] = b[i] = (1ull << 53) - 1;
        if (it == 1) for (int i = 0; i < 5; i++) { a[i] = 0; b[i] = 0; }
        uint32_t ha[10], hb[10], hc[10], hr[10], hu[10];
        H::from_limbs(a, ha); H::from_limbs(b, hb); H::from_limbs(c, hc);
        H::to_limbs(ha, got);
        int d = memcmp(got, a, sizeof got) != 0;                                  // the resident form is the same element
        H::modmul(ha, hb, hr); H::to_limbs(hr, got); modmul_X25519(a, b, want); d |= memcmp(got, want, sizeof got) != 0;
        H::modsqr(ha, hr); H::to_limbs(hr, got); modsqr_X25519(a, want); d |= memcmp(got, want, sizeof got) != 0;
        // the sums take field elements as the API defines them (SURVEY 8c caveat 2): any representative below 2p, limbs possibly over
        // Radix bits -- lower limbs up to 2^53, top limb below 2^52 - 8.  (For a VALUE of 2p or more the reference's modsub / modneg
        // leave a negative top limb, 64 bits wide there and 32 here: outside every contract of the API.)
        a[4] &= (1ull << 52) - 9; b[4] &= (1ull << 52) - 9;
        if (a[4] > (1ull << 52) - 9) a[4] = (1ull << 52) - 9;
        if (b[4] > (1ull << 52) - 9) b[4] = (1ull << 52) - 9;
        H::from_limbs(a, ha); H::from_limbs(b, hb);
        H::modadd(ha, hb, hr); H::to_limbs(hr, got); modadd_X25519(a, b, want); d |= memcmp(got, want, sizeof got) != 0;
        H::modsub(ha, hb, hr); H::to_limbs(hr, got); modsub_X25519(a, b, want); d |= memcmp(got, want, sizeof got) != 0;
        H::modneg(ha, hr); H::to_limbs(hr, got); modneg_X25519(a, want); d |= memcmp(got, want, sizeof got) != 0;
        // a product of sums, as the formulas chain them: (a + b)(a - b), squared
        H::modadd(ha, hb, hr); H::modsub(ha, hb, hu); H::modmul(hr, hu, hr); H::modsqr(hr, hr); H::to_limbs(hr, got);
        modadd_X25519(a, b, want); modsub_X25519(a, b, t); modmul_X25519(want, t, want); modsqr_X25519(want, want);
        d |= memcmp(got, want, sizeof got) != 0;
        // "_u" sums feeding sums: (a - b) - c, a - 2c, -a + b, -a - b, ((a + b) + c) - a
        H::modsub_u(ha, hb, hu); H::modsub(hu, hc, hr); H::to_limbs(hr, got); modsub_X25519(a, b, t); modsub_X25519(t, c, want); d |= memcmp(got, want, sizeof got) != 0;
        H::modadd_u(hc, hc, hu); H::modsub(ha, hu, hr); H::to_limbs(hr, got); modadd_X25519(c, c, t); modsub_X25519(a, t, want); d |= memcmp(got, want, sizeof got) != 0;
        H::modneg_u(ha, hu); H::modadd(hu, hb, hr); H::to_limbs(hr, got); modneg_X25519(a, t); modadd_X25519(t, b, want); d |= memcmp(got, want, sizeof got) != 0;
        H::modsub(hu, hb, hr); H::to_limbs(hr, got); modsub_X25519(t, b, want); d |= memcmp(got, want, sizeof got) != 0;
        H::modadd_u(ha, hb, hu); H::modadd_u(hu, hc, hu); H::modsub(hu, ha, hr); H::to_limbs(hr, got);
        modadd_X25519(a, b, t); modadd_X25519(t, c, t); modsub_X25519(t, a, want); d |= memcmp(got, want, sizeof got) != 0;
        if (d) { if (bad < 4) printf("FieldH51: record %d differs\n", it); bad++; }
    }
    printf("FieldH51<P_X25519> resident half-limb field (mul sqr add sub neg, _u chains): %d records, %d differ from the oracle\n", n, bad);
    return bad;
}


extern "C" void modmli_X448(const uint64_t*, int, uint64_t*);
extern "C" void modone_X448(uint64_t*);
// FieldH56<P_X448>: the resident half-limb form of the 8 x 56-bit Montgomery field against the oracle's limb functions
static int run_fh56(int n) {
    using Hf = ma::FieldH56<ma::P_X448>;
    int bad = 0;
    const uint64_t R = 56, Q = 1ull << R;
    const uint64_t edge[] = {0, 1, Q - 1, Q, 2 * Q - 1, 4 * Q - 1, (1ull << 28) - 1, 1ull << 28, Q - 2, 2, 3, (1ull << 29) - 1, Q - 3};
    for (int it = 0; it < n; it++) {
        uint64_t a[8], b[8], c[8], got[8], want[8], t[8];
        for (int i = 0; i < 8; i++) {
            uint64_t r = sm();
            a[i] = (r % 10 < 8 && it % 3) ? edge[r % 13] : (sm() & (4 * Q - 1));
            r = sm();
            b[i] = (r % 10 < 8 && it % 3 == 1) ? edge[r % 13] : (sm() & (4 * Q - 1));
            c[i] = sm() & (Q - 1);
        }
        if (it == 0) for (int i = 0; i < 8; i++) a[i] = b[i] = 4 * Q - 1;
        if (it == 1) for (int i = 0; i < 8; i++) { a[i] = 0; b[i] = 0; }
        uint32_t ha[16], hb[16], hc[16], hr[16], hu[16];
        Hf::from_limbs(a, ha); Hf::from_limbs(b, hb); Hf::from_limbs(c, hc);
        Hf::to_limbs(ha, got);
        int d = 0;                                                                // the resident form is the same element: same integer,
        { unsigned __int128 cg = 0, ca = 0;                                      // and the same limbs when none exceeds 56 bits
          bool same = true, tight = true;
          for (int i = 0; i < 8; i++) {
              cg += got[i]; ca += a[i];
              if (i < 7) { same = same && (uint64_t)(cg & (Q - 1)) == (uint64_t)(ca & (Q - 1)); cg >>= 56; ca >>= 56; tight = tight && a[i] < Q; }
          }
          same = same && cg == ca;
          d |= !same || (tight && memcmp(got, a, sizeof got) != 0); }
        Hf::modmul(ha, hb, hr); Hf::to_limbs(hr, got); modmul_X448(a, b, want); d |= memcmp(got, want, sizeof got) != 0;
        Hf::modsqr(ha, hr); Hf::to_limbs(hr, got); modsqr_X448(a, want); d |= memcmp(got, want, sizeof got) != 0;
        const int ml = 2 + (int)(sm() % 40000);
        Hf::modmli(ha, ml, hr); Hf::to_limbs(hr, got); modmli_X448(a, ml, want); d |= memcmp(got, want, sizeof got) != 0;
        Hf::modmli(ha, 39081, hr); Hf::to_limbs(hr, got); modmli_X448(a, 39081, want); d |= memcmp(got, want, sizeof got) != 0;
        // the sums take field elements as the API defines them: any representative below 2p, limbs possibly over Radix bits -- lower
        // limbs up to 2^58, top limb below 2^57 - 8
        a[7] &= (2 * Q) - 9; b[7] &= (2 * Q) - 9;
        Hf::from_limbs(a, ha); Hf::from_limbs(b, hb);
        Hf::modadd(ha, hb, hr); Hf::to_limbs(hr, got); modadd_X448(a, b, want); d |= memcmp(got, want, sizeof got) != 0;
        Hf::modsub(ha, hb, hr); Hf::to_limbs(hr, got); modsub_X448(a, b, want); d |= memcmp(got, want, sizeof got) != 0;
        Hf::modneg(ha, hr); Hf::to_limbs(hr, got); modneg_X448(a, want); d |= memcmp(got, want, sizeof got) != 0;
        // a product of sums, as the formulas chain them: (a + b)(a - b), squared, times 39081
        Hf::modadd(ha, hb, hr); Hf::modsub(ha, hb, hu); Hf::modmul(hr, hu, hr); Hf::modsqr(hr, hr); Hf::modmli(hr, 39081, hr); Hf::to_limbs(hr, got);
        modadd_X448(a, b, want); modsub_X448(a, b, t); modmul_X448(want, t, want); modsqr_X448(want, want); modmli_X448(want, 39081, want);
        d |= memcmp(got, want, sizeof got) != 0;
        // "_u" sums feeding sums: (a - b) - c, a - 2c, -a + b, -a - b, ((a + b) + c) - a
        Hf::modsub_u(ha, hb, hu); Hf::modsub(hu, hc, hr); Hf::to_limbs(hr, got); modsub_X448(a, b, t); modsub_X448(t, c, want); d |= memcmp(got, want, sizeof got) != 0;
        Hf::modadd_u(hc, hc, hu); Hf::modsub(ha, hu, hr); Hf::to_limbs(hr, got); modadd_X448(c, c, t); modsub_X448(a, t, want); d |= memcmp(got, want, sizeof got) != 0;
        Hf::modneg_u(ha, hu); Hf::modadd(hu, hb, hr); Hf::to_limbs(hr, got); modneg_X448(a, t); modadd_X448(t, b, want); d |= memcmp(got, want, sizeof got) != 0;
        Hf::modsub(hu, hb, hr); Hf::to_limbs(hr, got); modsub_X448(t, b, want); d |= memcmp(got, want, sizeof got) != 0;
        Hf::modadd_u(ha, hb, hu); Hf::modadd_u(hu, hc, hu); Hf::modsub(hu, ha, hr); Hf::to_limbs(hr, got);
        modadd_X448(a, b, t); modadd_X448(t, c, t); modsub_X448(t, a, want); d |= memcmp(got, want, sizeof got) != 0;
        if (it == 2) { Hf::modone(hr); Hf::to_limbs(hr, got); modone_X448(want); d |= memcmp(got, want, sizeof got) != 0; }
        if (d) { if (bad < 4) printf("FieldH56: record %d differs\n", it); bad++; }
    }
    printf("FieldH56<P_X448> resident half-limb field (mul sqr mli add sub neg, _u chains): %d records, %d differ from the oracle\n", n, bad);
    return bad;
}


template <class P, int N, class ADD, class SUB, class NEG>
static int run_u(const char* name, int n, int radix, ADD oadd, SUB osub, NEG oneg) {
    using F = ma::Field<P, true>;
    int bad = 0;
    for (int it = 0; it < n; it++) {
        uint64_t a[N], b[N], c[N], u[N], got[N], want[N], t[N];
        for (int i = 0; i < N; i++) {
            const uint64_t m = (1ull << (radix + (it % 4 == 0 ? 2 : 0))) - 1;       // every fourth record up to the contract's edge
            a[i] = (it % 7 == 3) ? ((sm() & 1) ? m : 0) : (sm() & m);
            b[i] = (it % 7 == 5) ? ((sm() & 1) ? m : 0) : (sm() & m);
            c[i] = sm() & ((1ull << radix) - 1);
        }
        int d = 0;
        F::modsub_u(a, b, u); F::modsub(u, c, got); osub(a, b, t); osub(t, c, want); d |= memcmp(got, want, sizeof got) != 0;
        F::modadd_u(c, c, u); F::modsub(a, u, got); oadd(c, c, t); osub(a, t, want); d |= memcmp(got, want, sizeof got) != 0;
        F::modneg_u(a, u); F::modadd(u, b, got); oneg(a, t); oadd(t, b, want); d |= memcmp(got, want, sizeof got) != 0;
        F::modsub(u, b, got); osub(t, b, want); d |= memcmp(got, want, sizeof got) != 0;
        F::modadd_u(a, b, u); F::modadd_u(u, c, u); F::modsub_u(u, a, u); F::modadd(u, u, got);
        oadd(a, b, t); oadd(t, c, t); osub(t, a, t); oadd(t, t, want); d |= memcmp(got, want, sizeof got) != 0;
        if (d) { if (bad < 4) printf("%s _u sums: record %d differs\n", name, it); bad++; }
    }
    printf("Field<P_%s,true> _u sums feeding sums: %d records, %d differ from the oracle\n", name, n, bad);
    return bad;
}

extern "C" void ecn_ed25519_add(pt25519*, pt25519*);
extern "C" void ecn_nist256_add(pt256*, pt256*);
extern "C" void ecn_nist256_dbl(pt256*);
extern "C" void ecn_ed448_add(pt448*, pt448*);
extern "C" void ecn_ed448_dbl(pt448*);

// E = the curve class under test (its add / dbl on its resident field form), PT = the oracle's point struct with N limbs per coordinate
template <class E, class PT, int N, int NBYTES, class GEN, class MUL, class ADD, class DBL>
static int run_formulas(const char* name, int n, GEN gen, MUL mul, ADD oadd, DBL odbl) {
    using F = typename E::F;
    int bad = 0;
    for (int it = 0; it < n; it++) {
        PT P, Q;
        char e[NBYTES], f[NBYTES];
        for (int i = 0; i < NBYTES; i++) { e[i] = (char)sm(); f[i] = (char)sm(); }
        if (it == 1) memset(f, 0, NBYTES);                     // Q = the neutral element
        gen(&P); mul(e, &P); gen(&Q); mul(f, &Q);
        if (it == 2) Q = P;                                     // P + P through the addition formula
        typename E::Point p, q;
        F::from_limbs(P.x, p.x); F::from_limbs(P.y, p.y); F::from_limbs(P.z, p.z);
        F::from_limbs(Q.x, q.x); F::from_limbs(Q.y, q.y); F::from_limbs(Q.z, q.z);
        PT W = P, G;
        odbl(&W);
        typename E::Point d;
        E::cpy(p, d); E::dbl(d);
        F::to_limbs(d.x, G.x); F::to_limbs(d.y, G.y); F::to_limbs(d.z, G.z);
        int df = memcmp(&G, &W, sizeof G) != 0;
        W = P; oadd(&Q, &W);                                    // P += Q
        E::cpy(p, d); E::add(q, d);
        F::to_limbs(d.x, G.x); F::to_limbs(d.y, G.y); F::to_limbs(d.z, G.z);
        df |= memcmp(&G, &W, sizeof G) != 0;
        if (df) { if (bad < 4) printf("%s: record %d differs\n", name, it); bad++; }
    }
    printf("%s add / dbl (projective limbs): %d records, %d differ from the oracle\n", name, n, bad);
    return bad;
}


// Fe28::sub at the limb extremes (round 4).  Its operands may each be a sum of two tight values; the subtrahend's limb can then sit
// at 2^29 (+ 2^10 for limbs 1 and 9) while the minuend's is 0 -- a pattern no random input has, and the doubling of the order-4
// point (0, 1, p-1) does.  Checked without the oracle: r + g == f (mod p) on the canonical words, and r is tight.
static int run_fe28_sub_extremes(int n) {
    using F = ma::Fe28;
    int bad = 0;
    uint64_t s = 0x9e3779b97f4a7c15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    auto tight = [&](uint32_t* t) {
        for (int i = 0; i < 16; i++) {
            const uint64_t r = rnd();
            const uint32_t hi = (i == 1 || i == 9) ? (1u << 28) + (1u << 9) - 1 : (1u << 28) - 1;
            switch (r & 7) {
                case 0: t[i] = 0; break;
                case 1: t[i] = 1; break;
                case 2: t[i] = hi; break;
                case 3: t[i] = hi - 1; break;
                case 4: t[i] = (1u << 28) - 1; break;
                default: t[i] = (uint32_t)(r >> 8) % (hi + 1);
            }
        }
    };
    for (int it = 0; it < n; it++) {
        uint32_t a[16], b[16], c[16], d[16], f[16], g[16], r[16], back[16];
        tight(a); tight(b); tight(c); tight(d);
        const uint64_t m = rnd();
        for (int i = 0; i < 16; i++) { f[i] = a[i] + ((m & 1) ? b[i] : 0); g[i] = c[i] + ((m & 2) ? d[i] : 0); }
        if (it % 4 == 0) for (int i = 0; i < 16; i++) { f[i] = (m >> (8 + i)) & 1; g[i] = ((i == 1 || i == 9) ? (1u << 29) + (1u << 10) - 2 : (1u << 29) - 2); }
        F::sub(f, g, r);
        bool ok = true;
        for (int i = 0; i < 16; i++) ok = ok && r[i] < (1u << 28) + ((i == 1 || i == 9) ? (1u << 9) : 0);
        F::add(r, g, back);
        uint64_t w1[7], w2[7];
        F::to_words(back, w1);
        F::to_words(f, w2);
        for (int k = 0; k < 7; k++) ok = ok && w1[k] == w2[k];
        if (!ok) { if (bad < 5) printf("Fe28::sub extremes: record %d differs\n", it); bad++; }
    }
    printf("Fe28::sub at the limb extremes: %d records, %d differ from f - g (mod p)\n", n, bad);
    return bad;
}

int main(int argc, char** argv) {
    int n = argc > 1 ? atoi(argv[1]) : 2000;
    int bad = run<4>("x25519_fe26_one", n, [](const uint64_t* k, const uint64_t* u, uint64_t* o) { ma::x25519_fe26_one(k, u, o); }, rfc7748_X25519);
    bad += run<7>("x448_fe28_one", n / 4 + 8, [](const uint64_t* k, const uint64_t* u, uint64_t* o) { ma::x448_fe28_one(k, u, o); }, rfc7748_X448);
    bad += run_fe28_sub_extremes(n * 50);
    bad += run_fh51(n * 50);
    bad += run_fh56(n * 25);
    bad += run_u<ma::P_X25519, 5>("X25519", n * 25, 51, modadd_X25519, modsub_X25519, modneg_X25519);
    bad += run_u<ma::P_NIST256, 5>("NIST256", n * 25, 52, modadd_NIST256, modsub_NIST256, modneg_NIST256);
    bad += run_u<ma::P_X448, 8>("X448", n * 25, 56, modadd_X448, modsub_X448, modneg_X448);
    bad += run_formulas<ma::Edwards<ma::C_ED25519, ma::FieldH51<ma::P_X25519>>, pt25519, 5, 32>("Edwards<ED25519> on FieldH51", n / 8 + 24, ecn_ed25519_gen, ecn_ed25519_mul, ecn_ed25519_add, ecn_ed25519_dbl);
    bad += run_formulas<ma::Edwards<ma::C_ED25519>, pt25519, 5, 32>("Edwards<ED25519> on limbs", n / 8 + 24, ecn_ed25519_gen, ecn_ed25519_mul, ecn_ed25519_add, ecn_ed25519_dbl);
    bad += run_formulas<ma::Edwards<ma::C_ED448>, pt448, 8, 56>("Edwards<ED448>", n / 16 + 24, ecn_ed448_gen, ecn_ed448_mul, ecn_ed448_add, ecn_ed448_dbl);
    bad += run_formulas<ma::Edwards<ma::C_ED448, ma::FieldH56<ma::P_X448>>, pt448, 8, 56>("Edwards<ED448> on FieldH56", n / 16 + 24, ecn_ed448_gen, ecn_ed448_mul, ecn_ed448_add, ecn_ed448_dbl);
    bad += run_formulas<ma::Weierstrass<ma::C_NIST256>, pt256, 5, 32>("Weierstrass<NIST256>", n / 8 + 24, ecn_nist256_gen, ecn_nist256_mul, ecn_nist256_add, ecn_nist256_dbl);
    bad += run_half(n * 50);
    bad += run_split4(n * 10);
    bad += run_mhalf(n * 50);
    bad += run_mhalf448(n * 25);
    bad += run_half_ov(n * 50);
    bad += run_fold52(n * 50);
    bad += run_ed25519_mul2("ed25519_mul2_get_straus_one", n / 4 + 32, [](const uint64_t* ew, const uint64_t* PX, const uint64_t* PY, const uint64_t* PZ, const uint64_t* fw, const uint64_t* QX, const uint64_t* QY, const uint64_t* QZ, uint64_t* xw, uint64_t* yw) { ma::ed25519_mul2_get_straus_one<ma::C_ED25519>(ew, PX, PY, PZ, fw, QX, QY, QZ, xw, yw); });
    bad += run_ed25519_lad("Ed26Lad::mul_get_one", n / 4 + 16, [](const uint64_t* ew, const uint64_t* X, const uint64_t* Y, const uint64_t* Z, uint64_t* xw, uint64_t* yw) { ma::Ed26Lad<ma::C_ED25519>::mul_get_one(ew, X, Y, Z, xw, yw); });
    bad += run_ed448_lad(n / 16 + 16);
    bad += run_ed448_mul2(n / 16 + 24);
    bad += run_ed448_mulgen2(n / 16 + 24);
    bad += run_nist256(n / 8 + 16);
    bad += run_nist256_mul2(n / 16 + 16);
    bad += run<4>("x25519_base_many<4> (u = 9)", n, [](const uint64_t* k, const uint64_t*, uint64_t* o) {
                      uint64_t ow[4][4];          // this scalar as element 1 of a group of four sharing the inversion
                      ma::x25519_base_many<ma::C_ED25519, HostComb25519, 4>([&](int g, uint64_t* kw) { for (int i = 0; i < 4; i++) kw[i] = g == 1 ? k[i] : (0x0123456789abcdefull * (g + 1)) ^ k[(i + g) & 3]; }, ow);
                      for (int i = 0; i < 4; i++) o[i] = ow[1][i];
                  }, rfc7748_X25519, 9);
    bad += run<7>("x448_base_one (u = 5)", n / 4 + 8, [](const uint64_t* k, const uint64_t*, uint64_t* o) { ma::x448_base_one<HostComb448>(k, o); }, rfc7748_X448, 5);
    bad += run_edgen<32, pt25519>("ed25519_mulgen_get_one", n / 4 + 80, [](const uint64_t* e, uint64_t* x, uint64_t* y) { ma::ed25519_mulgen_get_one<ma::C_ED25519, HostComb25519>(e, x, y); },
                                  ecn_ed25519_gen, ecn_ed25519_mul, ecn_ed25519_get);
    bad += run_ed25519_mulgen2_lad(n / 8 + 24);
    bad += run_edgen<56, pt448>("ed448_mulgen_get_one", n / 8 + 130, [](const uint64_t* e, uint64_t* x, uint64_t* y) { ma::ed448_mulgen_get_one<HostComb448>(e, x, y); },
                                ecn_ed448_gen, ecn_ed448_mul, ecn_ed448_get);
    return bad ? 1 : 0;
}
