// modarith_amd/csrc/fk26.h -- GF(p), p = 2^256 - 2^32 - 977 (the secp256k1 field), in ten SIGNED 26-bit limbs for the
// fused secp256k1 scalar multiplication (csrc/wn26.h) on gfx950.  Same idea as fm26.h; this prime is pseudo-Mersenne
// (pseudo.py's "overflow" form, 5 x 52 bits, no Montgomery: pseudo.py:368-416, 1641-1648), so the value is kept plain and
// the reduction is a fold:  B^10 = 2^260 = 2^36 + 0x3d10 (mod p),  B = 2^26.
//   T = sum_k acc_k B^k (19 columns).  The high columns 10..18 are carried into 26-bit digits h_0..h_8 and a rest h_9
//   first; digit h_k then joins column k as h_k * 0x3d10 and column k+1 as h_k * 2^10; h_9 reaches column 10 and folds
//   once more (columns 0 and 1).  The carry out of column 9 wraps the same way into limbs 0..2.
// 100 + 23 multiply-adds per multiplication, 55 + 23 per squaring; additions and subtractions are ten 32-bit operations
// without reduction.  Outputs have limbs in [0, 2^26) (limb 2 up to 2^26 + 2^18): "K = 1"; a lazy value has |limb| <= K 2^26 and
// a product needs K_f K_g <= 30 (the rest h_9 must fit a 32-bit multiplier).  wn26.h states K at every step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "field.h"

namespace ma {

struct Fk26 {
    static constexpr int32_t M26 = (1 << 26) - 1;
    static constexpr int32_t FC = 0x3d10;             // 2^260 mod p = 2^36 + FC
    static constexpr int32_t prime(int i) {           // canonical digits of p
        return i == 0 ? M26 - 976 : i == 1 ? M26 - 64 : i < 9 ? M26 : 0x3fffff;
    }

    // Multiplicands as values the compiler knows nothing about.  Where it can prove a limb non-negative it turns the sign
    // extension of the 64-bit product into a zero extension; a signed x zero-extended product has no single instruction
    // (v_mad_i64_i32 wants two sign extensions, v_mad_u64_u32 two zero extensions) and becomes two multiply-adds plus moves:
    // measured 1 959 instead of 1 422 multiply-adds in the addition of the main loop.  No instruction is emitted for this.
    static MA_DEV void opaque(const int32_t* f, int32_t* r) {
        static_for<0, 10>([&](auto I) {
            int32_t x = f[I];
#if defined(__HIP_DEVICE_COMPILE__)
            asm("" : "+v"(x));
#endif
            r[I] = x;
        });
    }
    // MODE 0: r = f g;  MODE 1: r = f^2;  MODE 3: r = f g + u v (one fold for both products; K_f K_g + K_u K_v <= 30)
    template <int MODE>
    static MA_DEV void prod(const int32_t* f, const int32_t* g, int32_t* r, const int32_t* u = nullptr, const int32_t* v = nullptr) {
        int32_t fo[10], go[10], uo[10], vo[10];
        opaque(f, fo);
        if constexpr (MODE != 1) opaque(g, go);
        if constexpr (MODE == 3) { opaque(u, uo); opaque(v, vo); }
        f = fo; g = go; u = uo; v = vo;
        int32_t f2[10];
        if constexpr (MODE == 1) static_for<0, 10>([&](auto I) { f2[I] = (int32_t)(2u * (uint32_t)f[I]); });
        // 2^10, 2^20 and the folding constants as opaque scalar registers (see fm26.h: constants would become 64-bit shifts + adds)
        int32_t c10 = 1 << 10, c20 = 1 << 20;
#if defined(__HIP_DEVICE_COMPILE__)
        asm("s_mov_b32 %0, 0x400" : "=s"(c10));
        asm("s_mov_b32 %0, 0x100000" : "=s"(c20));
#endif
        auto column = [&](auto KK, int64_t& acc) {
            constexpr int k = KK;
            static_for<0, 10>([&](auto II) {
                constexpr int i = II;
                constexpr int j = k - i;
                if constexpr (j >= 0 && j < 10) {
                    if constexpr (MODE == 0) {
                        acc += (int64_t)f[i] * g[j];
                        MA_PIN(acc);
                    } else if constexpr (MODE == 3) {
                        acc += (int64_t)f[i] * g[j];
                        MA_PIN(acc);
                        acc += (int64_t)u[i] * v[j];
                        MA_PIN(acc);
                    } else if constexpr (i < j) {
                        acc += (int64_t)f2[i] * f[j];
                        MA_PIN(acc);
                    } else if constexpr (i == j) {
                        acc += (int64_t)f[i] * f[i];
                        MA_PIN(acc);
                    }
                }
            });
        };
        int32_t h[10], t[10];
        int64_t c = 0;
        static_for<10, 19>([&](auto KK) {
            constexpr int k = KK;
            int64_t acc = c;
            column(KK, acc);
            h[k - 10] = (int32_t)((uint32_t)acc & (uint32_t)M26);
            c = acc >> 26;
        });
        h[9] = (int32_t)c;                              // |h_9| <= K_f K_g 2^26
        c = 0;
        static_for<0, 10>([&](auto KK) {
            constexpr int k = KK;
            int64_t acc = c;
            column(KK, acc);
            acc += (int64_t)h[k] * FC;
            MA_PIN(acc);
            if constexpr (k >= 1) { acc += (int64_t)h[k - 1] * c10; MA_PIN(acc); }
            if constexpr (k == 0) { acc += (int64_t)h[9] * (FC << 10); MA_PIN(acc); }     // h_9 B^10 2^10 = h_9 2^10 (2^10 B + FC)
            if constexpr (k == 1) { acc += (int64_t)h[9] * c20; MA_PIN(acc); }
            t[k] = (int32_t)((uint32_t)acc & (uint32_t)M26);
            c = acc >> 26;
        });
        // the carry out of column 9 (|c| < 2^35) wraps: c B^10 = c FC + c 2^10 B
        const int64_t a0 = (int64_t)t[0] + c * FC;
        t[0] = (int32_t)((uint32_t)a0 & (uint32_t)M26);
        const int64_t a1 = (int64_t)t[1] + c * (int64_t)(1 << 10) + (a0 >> 26);
        t[1] = (int32_t)((uint32_t)a1 & (uint32_t)M26);
        t[2] += (int32_t)(a1 >> 26);
        static_for<0, 10>([&](auto I) { r[I] = t[I]; });
    }
    static MA_DEV void mul(const int32_t* f, const int32_t* g, int32_t* r) { prod<0>(f, g, r); }
    static MA_DEV void sqr(const int32_t* f, int32_t* r) { prod<1>(f, f, r); }
    static MA_DEV void mul2(const int32_t* f, const int32_t* g, const int32_t* u, const int32_t* v, int32_t* r) { prod<3>(f, g, r, u, v); }
    // r = f * S for a small positive constant (3b = 21), carried: K = 1 whatever K_f (|f| S < 2^31 B)
    template <int32_t S>
    static MA_DEV void mul_small(const int32_t* f, int32_t* r) {
        int64_t c = 0;
        int32_t t[10];
        static_for<0, 10>([&](auto KK) {
            constexpr int k = KK;
            const int64_t acc = c + (int64_t)f[k] * S;
            t[k] = (int32_t)((uint32_t)acc & (uint32_t)M26);
            c = acc >> 26;
        });
        const int64_t a0 = (int64_t)t[0] + c * FC;       // |c| < 2^10
        t[0] = (int32_t)((uint32_t)a0 & (uint32_t)M26);
        const int64_t a1 = (int64_t)t[1] + c * (int64_t)(1 << 10) + (a0 >> 26);
        t[1] = (int32_t)((uint32_t)a1 & (uint32_t)M26);
        t[2] += (int32_t)(a1 >> 26);
        static_for<0, 10>([&](auto I) { r[I] = t[I]; });
    }

    // limb-wise, in WRAPPING 32-bit arithmetic: with the signed operators (overflow undefined) the compiler is entitled to do
    // the addition in 64 bits after sign extension, and then multiplies the 64-bit sum with two v_mad_u64_u32 plus moves
    // instead of one v_mad_i64_i32 (measured: 1 959 instead of 1 422 multiply-adds in the secp256k1 addition)
    static MA_DEV void add(const int32_t* f, const int32_t* g, int32_t* r) { static_for<0, 10>([&](auto I) { r[I] = (int32_t)((uint32_t)f[I] + (uint32_t)g[I]); }); }
    static MA_DEV void sub(const int32_t* f, const int32_t* g, int32_t* r) { static_for<0, 10>([&](auto I) { r[I] = (int32_t)((uint32_t)f[I] - (uint32_t)g[I]); }); }
    static MA_DEV void neg(const int32_t* f, int32_t* r) { static_for<0, 10>([&](auto I) { r[I] = (int32_t)(0u - (uint32_t)f[I]); }); }
    static MA_DEV void copy(const int32_t* f, int32_t* r) { static_for<0, 10>([&](auto I) { r[I] = f[I]; }); }
    static MA_DEV void zero(int32_t* r) { static_for<0, 10>([&](auto I) { r[I] = 0; }); }
    static MA_DEV void set_one(int32_t* r) { static_for<0, 10>([&](auto I) { r[I] = (I == 0) ? 1 : 0; }); }
    static MA_DEV void select(bool s, const int32_t* f, const int32_t* g, int32_t* r) {
        static_for<0, 10>([&](auto I) {
            const int32_t x = f[I], y = g[I];
            r[I] = s ? y : x;
        });
    }
    static MA_DEV void sqn(int32_t* f, int n) {
#pragma unroll 1
        for (int i = 0; i < n; i++) sqr(f, f);
    }

    // z^(p-2), p - 2 = 2^256 - 2^32 - 979: 223 ones, a zero, 22 ones, 0000101101: 255 squarings, 15 multiplications
    static MA_DEV void invert(const int32_t* z, int32_t* out) {
        int32_t x2[10], x3[10], x22[10], t[10], s[10];
        sqr(z, x2);    mul(x2, z, x2);                        // 2^2 - 1
        sqr(x2, x3);   mul(x3, z, x3);                        // 2^3 - 1
        copy(x3, t);   sqn(t, 3);   mul(t, x3, t);            // x6
        copy(t, s);    sqn(s, 3);   mul(s, x3, s);            // x9
        copy(s, t);    sqn(t, 2);   mul(t, x2, t);            // x11
        copy(t, x22);  sqn(x22, 11); mul(x22, t, x22);        // x22
        copy(x22, t);  sqn(t, 22);  mul(t, x22, t);           // x44
        copy(t, s);    sqn(s, 44);  mul(s, t, s);             // x88
        copy(s, out);  sqn(out, 88); mul(out, s, out);        // x176   (out used as scratch)
        sqn(out, 44);  mul(out, t, out);                      // x220
        sqn(out, 3);   mul(out, x3, out);                     // x223
        copy(out, t);
        sqn(t, 23);    mul(t, x22, t);
        sqn(t, 5);     mul(t, z, t);
        sqn(t, 3);     mul(t, x2, t);
        sqn(t, 2);     mul(t, z, out);
    }

    // field.c form (5 x 52-bit limbs of the plain value, limbs below 2^54: the contract of the curve layer) -> this form
    static MA_DEV void from52(const spint* x, int32_t* r) {
        static_for<0, 5>([&](auto K) {
            constexpr int k = K;
            r[2 * k] = (int32_t)((uint32_t)x[k] & (uint32_t)M26);
            r[2 * k + 1] = (int32_t)(x[k] >> 26);               // < 2^28
        });
        mul_small<1>(r, r);                                    // carried: K = 1 (the products of wn26.h need K_f K_g <= 30)
    }
    // the value mod p, canonical, as four little-endian 64-bit words; |limb| < 2^30 on entry
    static MA_DEV void to_words(const int32_t* f, uint64_t* w) {
        int64_t t[10];
        // + 1024 p: every limb positive whatever the sign of the input (digit 9 of p is 2^22 - 1)
        static_for<0, 10>([&](auto I) { t[I] = (int64_t)f[I] + (int64_t)1024 * prime(I); });
        auto carry_fold = [&]() {
            static_for<0, 9>([&](auto I) {
                t[I + 1] += t[I] >> 26;
                t[I] &= M26;
            });
            const int64_t q = t[9] >> 22;                       // multiples of 2^256 = 2^32 + 977 (mod p)
            t[9] &= (1 << 22) - 1;
            t[0] += q * 977;
            t[1] += q * 64;
        };
        carry_fold();                                          // q < 2^19
        carry_fold();                                          // q <= 1
        carry_fold();                                          // value < 2^256 now (q = 0 unless it was >= 2^256: then tiny)
        static_for<0, 9>([&](auto I) {
            t[I + 1] += t[I] >> 26;
            t[I] &= M26;
        });
        int32_t d[10], s[10];
        static_for<0, 10>([&](auto I) { d[I] = (int32_t)t[I]; });
        int32_t bw = 0;                                        // s = d - p with borrow
        static_for<0, 10>([&](auto I) {
            constexpr int i = I;
            const int32_t x = d[i] - prime(i) + bw;
            bw = x >> 31;
            s[i] = (i < 9) ? (x & M26) : x;
        });
        const bool ge = bw == 0;
        static_for<0, 10>([&](auto I) { d[I] = ge ? s[I] : d[I]; });
        static_for<0, 4>([&](auto K) { w[K] = 0; });
        static_for<0, 10>([&](auto I) {
            constexpr int i = I;
            constexpr int o = 26 * i, wi = o / 64, sh = o % 64;
            w[wi] |= (uint64_t)(uint32_t)d[i] << sh;
            if constexpr (sh + 26 > 64 && wi + 1 < 4) w[wi + 1] |= (uint64_t)(uint32_t)d[i] >> (64 - sh);
        });
    }
    static MA_DEV void pack(const int32_t* f, uint64_t* w) {
        static_for<0, 5>([&](auto K) { w[K] = (uint64_t)(uint32_t)f[2 * K] | ((uint64_t)(uint32_t)f[2 * K + 1] << 32); });
    }
    static MA_DEV void unpack(const uint64_t* w, int32_t* f) {
        static_for<0, 5>([&](auto K) {
            int32_t lo = (int32_t)(uint32_t)w[K], hi = (int32_t)(uint32_t)(w[K] >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
            // cut the provenance: seen as "the high half of a 64-bit word" the compiler keeps hi as a sign-extended 64-bit value
            // and multiplies it with TWO v_mad_u64_u32 plus moves (a 64 x 32 product) instead of one v_mad_i64_i32
            asm("" : "+v"(lo));
            asm("" : "+v"(hi));
#endif
            f[2 * K] = lo;
            f[2 * K + 1] = hi;
        });
    }
};

}  // namespace ma
