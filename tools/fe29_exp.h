// tools/fe29_exp.h -- EXPERIMENT (round 3, not product code): GF(2^255-19) in nine 29-bit limbs, the 32-bit limb choice
// of the reference (simd/pseudo_cuda.py:1328: 9 x 29, 81 products per multiplication), against csrc/fe26.h's ten
// 25.5-bit limbs (100 products).  2^261 = 2^6 * 2^255 = 1216 (mod p) does not fit next to a 30-bit limb in a 32-bit
// multiplier operand, so the high columns cannot be pre-multiplied: they are summed apart, carried into 29-bit digits
// and each digit folded with ONE multiply-add by 1216 (the reference's "overflow" row form, pseudo.py:407-436).
// Per multiplication: 81 + 9 + 1 multiply-adds and 17 mask/shift pairs instead of 100 + 1 multiply-adds, 9 + 5
// premultiplications and 10 mask/shift pairs.
#pragma once
#include "../modarith_amd/csrc/field.h"

namespace ma {

struct Fe29 {
    static constexpr uint32_t M29 = (1u << 29) - 1;
    static constexpr uint32_t FOLD = 1216;                       // 2^261 mod p
    // value = sum f_i 2^(29 i).  tight: f_i < 2^29 (f_1 < 2^29 + 2^18).  mul accepts one operand up to 1.5 * 2^30 (a
    // 2p-biased difference) against one up to 2^30 (a sum): 9 * 1.5 * 2^60 + carries < 2^64; sqr accepts up to 2^30.
    static MA_DEV void wrap(uint64_t c, uint32_t* r) {
        uint64_t h0 = (uint64_t)r[0] + (uint64_t)FOLD * c;       // c < 2^36
        r[0] = (uint32_t)h0 & M29;
        r[1] += (uint32_t)(h0 >> 29);
    }
    static MA_DEV void mul(const uint32_t* f, const uint32_t* g, uint32_t* r) {
        uint64_t hc = 0, tc = 0;
        uint32_t t[9];
        static_for<0, 9>([&](auto KK) {
            constexpr int k = KK;
            uint32_t lo;
            if constexpr (k < 8) {
                uint64_t hacc = hc;
                static_for<k + 1, 9>([&](auto II) {
                    constexpr int i = II;
                    hacc += (uint64_t)f[i] * g[k + 9 - i];
                    MA_PIN(hacc);
                });
                lo = (uint32_t)hacc & M29;
                hc = hacc >> 29;
            } else {
                lo = (uint32_t)hc;                                // what is left above column 16: below 2^32
            }
            uint64_t acc = tc;
            static_for<0, k + 1>([&](auto II) {
                constexpr int i = II;
                acc += (uint64_t)f[i] * g[k - i];
                MA_PIN(acc);
            });
            acc += (uint64_t)lo * FOLD;
            MA_PIN(acc);
            t[k] = (uint32_t)acc & M29;
            tc = acc >> 29;
        });
        static_for<0, 9>([&](auto I) { r[I] = t[I]; });
        wrap(tc, r);
    }
    static MA_DEV void sqr(const uint32_t* f, uint32_t* r) {
        uint32_t f2[9];
        static_for<0, 8>([&](auto I) { f2[I] = 2u * f[I]; });
        uint64_t hc = 0, tc = 0;
        uint32_t t[9];
        static_for<0, 9>([&](auto KK) {
            constexpr int k = KK;
            uint32_t lo;
            if constexpr (k < 8) {
                uint64_t hacc = hc;
                static_for<k + 1, 9>([&](auto II) {
                    constexpr int i = II, j = k + 9 - i;
                    if constexpr (i < j) { hacc += (uint64_t)f2[i] * f[j]; MA_PIN(hacc); }
                    else if constexpr (i == j) { hacc += (uint64_t)f[i] * f[j]; MA_PIN(hacc); }
                });
                lo = (uint32_t)hacc & M29;
                hc = hacc >> 29;
            } else {
                lo = (uint32_t)hc;
            }
            uint64_t acc = tc;
            static_for<0, k + 1>([&](auto II) {
                constexpr int i = II, j = k - i;
                if constexpr (i < j) { acc += (uint64_t)f2[i] * f[j]; MA_PIN(acc); }
                else if constexpr (i == j) { acc += (uint64_t)f[i] * f[j]; MA_PIN(acc); }
            });
            acc += (uint64_t)lo * FOLD;
            MA_PIN(acc);
            t[k] = (uint32_t)acc & M29;
            tc = acc >> 29;
        });
        static_for<0, 9>([&](auto I) { r[I] = t[I]; });
        wrap(tc, r);
    }
    static MA_DEV void add(const uint32_t* f, const uint32_t* g, uint32_t* r) {
        static_for<0, 9>([&](auto I) { r[I] = f[I] + g[I]; });
    }
    // r = f - g + 2p, 2p = 2^256 - 38 with every limb large enough for a tight g: (2^30 - 38*... ) see limbs below
    // 2p = sum b_i 2^(29 i): b_0 = 2^30 - 38, b_1..b_7 = 2^30 - 2, b_8 = 2^24 - 2
    static MA_DEV void sub(const uint32_t* f, const uint32_t* g, uint32_t* r) {
        static_for<0, 9>([&](auto I) {
            constexpr int i = I;
            constexpr uint32_t b = (i == 0) ? ((1u << 30) - 38u) : (i == 8 ? ((1u << 24) - 2u) : ((1u << 30) - 2u));
            r[i] = (f[i] + b) - g[i];
        });
    }
    // one parallel carry step: limbs of a 2p-biased difference (< 1.5 * 2^30) come back below 2^29 + 4, so that the value may
    // be squared (9 * (1.5 * 2^30)^2 does not fit 64 bits); the top limb's carry wraps with 2^261 = 1216
    static MA_DEV void tighten(uint32_t* f) {
        uint32_t c[9];
        static_for<0, 9>([&](auto I) { c[I] = f[I] >> 29; });
        static_for<1, 9>([&](auto I) { f[I] = (f[I] & M29) + c[I - 1]; });
        f[0] = (f[0] & M29) + FOLD * c[8];
    }
    static MA_DEV void select(bool s, const uint32_t* f, const uint32_t* g, uint32_t* r) {
        static_for<0, 9>([&](auto I) {
            const uint32_t x = f[I], y = g[I];
            r[I] = s ? y : x;
        });
    }
    template <uint32_t C>
    static MA_DEV void mul_small_add(const uint32_t* f, const uint32_t* a, uint32_t* r) {
        uint64_t c = 0;
        uint32_t t[9];
        static_for<0, 9>([&](auto KK) {
            constexpr int k = KK;
            const uint64_t acc = c + (uint64_t)f[k] * C;
            MA_PIN(acc);
            t[k] = (uint32_t)acc & M29;
            c = acc >> 29;
        });
        wrap(c, t);
        static_for<0, 9>([&](auto I) { r[I] = t[I] + a[I]; });
    }
    // one ladder step (the arithmetic of fe26.h's step on nine limbs; x1 tight)
    static MA_DEV void step(bool sw, const uint32_t* x1, uint32_t* x2, uint32_t* z2, uint32_t* x3, uint32_t* z3) {
        uint32_t A[9], B[9], C[9], D[9], As[9], Bs[9], AA[9], BB[9], E[9];
        add(x2, z2, A);
        add(x3, z3, C);
        sub(x2, z2, B);
        sub(x3, z3, D);
        select(sw, A, C, As);
        select(sw, B, D, Bs);
        mul(D, A, D);
        mul(B, C, C);
        tighten(Bs);
        sqr(As, AA);
        sqr(Bs, BB);
        sub(D, C, z3);
        add(D, C, x3);
        sub(AA, BB, E);
        mul_small_add<121665>(E, AA, z2);
        mul(E, z2, z2);
        tighten(z3);
        sqr(x3, x3);
        sqr(z3, z3);
        mul(z3, x1, z3);
        mul(AA, BB, x2);
    }
};

}  // namespace ma
