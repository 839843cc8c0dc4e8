// modarith_amd/csrc/ed28.h -- fused ecnXXXmul + ecnXXXget for ED448 on the fe28 representation (gfx950).
//
// Same construction as ed26.h (which see): only canonical affine bytes leave the kernel, so the arithmetic runs on
// sixteen 28-bit limbs (fe28.h: golden-ratio Karatsuba products) and on extended coordinates (X:Y:Z:T) with the
// Hisil-Wong-Carter-Dawson formulas for a = 1, which are complete on x^2 + y^2 = 1 - 39081 x^2 y^2 (a = 1 is a square,
// d = -39081 a non-square mod 2^448 - 2^224 - 1): the affine point the reference's edwards.c formulas reach, for every
// input point on the curve.  The coordinates arrive in Montgomery form (monty.py, R = 2^504): a projective point whose
// three coordinates carry the same factor R is the same point, so the limbs are simply read as integers mod p.
//
// 3-bit signed fixed windows like ed26.h: e' = e + sum_{i<150} 4*8^i < 8^150, digit_i = window_i(e') - 4.  The table
// {1,2,3,4}P -- affine, cached as canonical packed (x, y, 39081 x y), 4 x 3 x 56 bytes per lane -- does not fit the
// register file next to a 64-register point, so it lives in a caller-provided device workspace laid out
// [entry][word][lane] (8-byte coalesced accesses; 88 MB for the resident grid, inside the Infinity Cache); every lookup
// reads all four entries and selects with v_cndmask.
//
// Limb bounds (fe28.h): "tight" = below 2^28 (+2^9 on limbs 1, 9); add() of tight values stays below 2^29; sub() carries
// back to tight; mul_k / sqr_k (Karatsuba) want one tight operand and one below 2^29 / a tight operand; mul / sqr take
// anything below 2^29 (one operand up to 1.5 * 2^29: a folded column holds 38 products, 57 * 2^58 < 2^64).
#pragma once
#include "fe28.h"

namespace ma {

struct Ed28 {
    using F = Fe28;
    static constexpr uint32_t M28 = F::M28;
    static constexpr uint32_t D_ABS = 39081;       // d = -39081 (curve.py: CONSTANT_B of ED448)
    struct Ext { uint32_t X[16], Y[16], Z[16], T[16]; };

    // limbs below 2^31 -> tight, 32-bit registers only (the carry out of limb 15 re-enters at limbs 0 and 8)
    static MA_DEV void wc(uint32_t* f) {
        static_for<0, 15>([&](auto I) {
            constexpr int i = I;
            f[i + 1] += f[i] >> 28;
            f[i] &= M28;
        });
        const uint32_t top = f[15] >> 28;
        f[15] &= M28;
        const uint32_t h0 = f[0] + top, h8 = f[8] + top;
        f[0] = h0 & M28;
        f[1] += h0 >> 28;
        f[8] = h8 & M28;
        f[9] += h8 >> 28;
    }
    // r = 2p - f for tight f: below 2^29, not carried (limbs of 2p: 2^29-2, limb 8: 2^29-4)
    static MA_DEV void neg2p(const uint32_t* f, uint32_t* r) {
        static_for<0, 16>([&](auto I) {
            constexpr int i = I;
            constexpr uint32_t twop = (i == 8) ? 0x1ffffffcu : 0x1ffffffeu;
            r[i] = twop - f[i];
        });
    }
    // 8 x 56-bit limbs (field.c form, limbs below 2^58: the contract of the curve layer) -> tight fe28
    static MA_DEV void from56(const spint* x, uint32_t* r) {
        static_for<0, 8>([&](auto K) {
            constexpr int k = K;
            r[2 * k] = (uint32_t)x[k] & M28;
            r[2 * k + 1] = (uint32_t)(x[k] >> 28);
        });
        wc(r);
    }

    // P = 2P, dbl-2008-hwcd with a = 1: A = X^2, B = Y^2, C = 2Z^2, G = A + B, E = (X+Y)^2 - G, F = G - C, H = A - B;
    // X3 = E F, Y3 = G H, Z3 = F G, T3 = E H.  Input X, Y, Z tight; T is not read.
    template <bool WANT_T>
    static MA_DEV void dbl(Ext& p) { dbl(p, WANT_T); }
    // want_t is wave-uniform (a loop counter, never data): the window loop keeps ONE copy of the doubling in the
    // instruction stream -- three unrolled copies plus the addition are about 80 KB of code, more than the 64 KB
    // instruction cache, and the kernel then waits on instruction fetch for 40 % of its time (SQ_WAIT_ANY)
    static MA_DEV void dbl(Ext& p, bool want_t) {
        uint32_t A[16], B[16], Cc[16], S[16], G[16], E[16], Ff[16], H[16];
        F::sqr_k(p.X, A);
        F::sqr_k(p.Y, B);
        F::sqr_k(p.Z, Cc);
        F::add(p.X, p.Y, S);        // < 2^29
        F::sqr(S, S);
        F::add(A, B, G);            // < 2^29
        F::sub(S, G, E);            // tight
        F::add(Cc, Cc, Cc);         // < 2^29
        F::sub(G, Cc, Ff);          // tight
        F::sub(A, B, H);            // tight
        F::mul_k(E, Ff, p.X);
        F::mul_k(H, G, p.Y);
        F::mul_k(Ff, G, p.Z);
        if (want_t) F::mul_k(E, H, p.T);
    }
    // common tail of the additions (a = 1, d = -39081): with A = X1 X2, B = Y1 Y2, Cc = 39081 T1 T2 (= -C), D = Z1 Z2,
    // M = (X1+Y1)(X2+Y2):  E = M - A - B, F = D - C = D + Cc, G = D + C = D - Cc, H = B - A;  X3 = E F, Y3 = G H, Z3 = F G
    static MA_DEV void add_tail(const uint32_t* A, const uint32_t* B, const uint32_t* Cc, const uint32_t* D, const uint32_t* M, Ext& p,
                                bool want_t = false) {
        uint32_t E[16], Ff[16], G[16], H[16], AB[16];
        F::add(A, B, AB);           // < 2^29
        F::sub(M, AB, E);           // tight
        F::add(D, Cc, Ff);          // < 2^29
        F::sub(D, Cc, G);           // tight
        F::sub(B, A, H);            // tight
        F::mul_k(E, Ff, p.X);
        F::mul_k(G, H, p.Y);
        F::mul_k(G, Ff, p.Z);
        if (want_t) F::mul_k(E, H, p.T);    // (wave-uniform flag: one copy of the addition serves both uses)
    }
    // P += Q, Q affine and cached as (x, y, td = 39081 x y), already sign-adjusted: xs, tds below 2^29, y tight.
    // Reads T of P; T of the sum is not produced (a doubling follows).
    static MA_DEV void add_cached(Ext& p, const uint32_t* xs, const uint32_t* y, const uint32_t* tds, bool want_t = false) {
        uint32_t A[16], B[16], Cc[16], M[16], s1[16], s2[16];
        F::mul_k(p.X, xs, A);
        F::mul_k(p.Y, y, B);
        F::mul_k(p.T, tds, Cc);
        F::add(p.X, p.Y, s1);       // < 2^29
        F::add(xs, y, s2);          // < 1.5 * 2^29
        F::mul(s1, s2, M);
        add_tail(A, B, Cc, p.Z, M, p, want_t);
    }
};

// (Rounds 2-4 kept here the window forms of the fused multiplications -- 3-bit / 2-bit signed windows over affine tables {1..4}P in a
// workspace slab, scanned; the one-point-at-a-time table builder Ed448Slots; the digit sources Win3 / Win2 -- 2.0e7 mul_get/s,
// 1.4e7 mul2_get/s.  Round 5 replaced them by the ladder form (ed28l.h) and the Straus form (ed28s.h); what is left below is the
// fixed-base part, which both still use.)

// Fused GENERATOR multiplication + affine export for ED448 (see ed26.h ed25519_mulgen_get_one): ED448_KEY_PAIR and
// ED448_SIGN open with ecnXXXgen, ecnXXXmul, ecnXXXget (ed448.c:167-184, 196-199).  W = 4: e' = e + sum_{i<113} 8*16^i, 113
// signed 4-bit digits, the 113 x 8 multiples m * 16^i * G precomputed as (x, y, 39081 x y) in sixteen 28-bit limbs
// (generated/comb_ED448.h, 173 568 bytes, wave-uniform reads; W = 5 measured the same), one complete mixed addition per window, no doublings.
template <class TAB, bool INIT = true>         // INIT = false: R += e*G (R holds a sum with its T coordinate)
MA_DEV void ed448_mulgen_acc(const uint64_t* ew, Ed28::Ext& R) {
    using E = Ed28;
    using F = Fe28;
    constexpr int W = TAB::W, NW = TAB::NW, E2 = 1 << (W - 1);       // window width, windows, entries per window
    static_assert(W * NW >= 449 && W * NW <= 512, "e + bias must fit the windows and eight words");
    uint64_t w[8];
    {
        constexpr auto cw = [](int k) {
            uint64_t v = 0;
            for (int b = 0; b < 64; b++) {
                const int pos = 64 * k + b;
                if (pos < W * NW && pos % W == W - 1) v |= (uint64_t)1 << b;
            }
            return v;
        };
        unsigned __int128 acc = 0;
        static_for<0, 8>([&](auto K) {
            constexpr int k = K;
            acc += (unsigned __int128)(k < 7 ? ew[k < 7 ? k : 0] : 0) + cw(k);
            w[k] = (uint64_t)acc;
            acc >>= 64;
        });
    }
    if constexpr (INIT) {
        F::set(0, R.X);
        F::set(1, R.Y);
        F::set(1, R.Z);
        F::set(0, R.T);
    }
#pragma unroll 1
    for (int i = 0; i < NW; i++) {
        const int dgt = (int)((uint32_t)w[0] & (uint32_t)(2 * E2 - 1)) - E2;        // [-2^(W-1), 2^(W-1) - 1]
        static_for<0, 8>([&](auto K) {
            constexpr int k = K;
            w[k] >>= W;
            if constexpr (k < 7) w[k] |= w[k + 1] << (64 - W);
        });
        const bool neg = dgt < 0;
        const uint32_t m = (uint32_t)(neg ? -dgt : dgt);        // 0 .. 2^(W-1)
        uint32_t sel[3][16];
        static_for<0, 3>([&](auto CI) { static_for<0, 16>([&](auto K) { sel[CI][K] = (CI == 1 && K == 0) ? 1u : 0u; }); });
        // (selection as OR of masked entries, see wn26.h; limb 0 of y starts at 1 only for a zero digit)
        sel[1][0] = (m == 0) ? 1u : 0u;
        static_for<0, E2>([&](auto MM) {
            constexpr int mm = MM;
            uint32_t mask = (m == (uint32_t)(mm + 1)) ? 0xffffffffu : 0u;
#if defined(__HIP_DEVICE_COMPILE__)
            asm("" : "+v"(mask));       // opaque: otherwise the compiler turns (entry & mask) back into a select with a move
#endif
            static_for<0, 3>([&](auto CI) {
                static_for<0, 16>([&](auto K) { sel[CI][K] |= (uint32_t)TAB::get(((i * E2 + mm) * 3 + CI) * 16 + K) & mask; });
            });
        });
        uint32_t nx[16], nt[16];
        E::neg2p(sel[0], nx);
        E::neg2p(sel[2], nt);
        F::select(neg, sel[0], nx, sel[0]);
        F::select(neg, sel[2], nt, sel[2]);
        E::add_cached(R, sel[0], sel[1], sel[2], true);
    }
}
template <class TAB>
MA_DEV void ed448_mulgen_get_one(const uint64_t* ew, uint64_t* xw, uint64_t* yw) {
    using F = Fe28;
    Ed28::Ext R;
    ed448_mulgen_acc<TAB>(ew, R);
    uint32_t zi[16], ax[16], ay[16];
    F::invert(R.Z, zi);
    F::mul_k(R.X, zi, ax);
    F::mul_k(R.Y, zi, ay);
    F::to_words(ax, xw);
    F::to_words(ay, yw);
}
// rfc7748() on the BASE POINT u = 5 of X448 (public-key generation, rfc7748.c:297-333).  ED448 (x^2 + y^2 = 1 - 39081 x^2 y^2) is
// 4-isogenous to curve448 with (u, v) = (y^2 / x^2, ...) (RFC 7748 section 4.2), and its generator maps to u = 5: the isogeny is
// a group homomorphism, so [k](5) = Y^2 / X^2 of k*G, k*G from the fixed-base table (ed448_mulgen_acc).  k clamped as
// rfc7748.c:135-141; k = 4q gives the neutral element, X = 0, and 0^(p-2) = 0 makes the result 0 as the ladder's.
template <class TAB>
MA_DEV void x448_base_one(const uint64_t* kw_in, uint64_t* ow) {
    using F = Fe28;
    uint64_t kw[7];
    static_for<0, 7>([&](auto K) { kw[K] = kw_in[K]; });
    kw[0] &= ~3ull;
    kw[6] |= 0x8000000000000000ull;
    Ed28::Ext R;
    ed448_mulgen_acc<TAB>(kw, R);
    uint32_t x2[16], y2[16], xi[16], u[16];
    F::sqr_k(R.X, x2);
    F::sqr_k(R.Y, y2);
    F::invert(x2, xi);
    F::mul_k(y2, xi, u);
    F::to_words(u, ow);
}

// (Rounds 2-4 ran TWO scalars per lane with one inversion -- ed448_mulgen_get_two / x448_base_two, the first result parked in LDS -- and
// spilled 14 / 2 registers in that epilogue; round 5 shares the inversion between up to 32 records instead: csrc/edlad_k.h,
// capi_ED448G.hip k_ed448_mulgen / k_x448_base.)

}  // namespace ma
