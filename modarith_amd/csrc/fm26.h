// modarith_amd/csrc/fm26.h -- GF(p256), p = 2^256 - 2^224 + 2^192 + 2^96 - 1, in ten SIGNED 26-bit limbs and Montgomery
// form with R' = 2^286, for the fused NIST P-256 scalar multiplication (csrc/wn26.h) on gfx950.
//
// Why a second representation (as fe26.h for 2^255-19): the fused kernels emit canonical bytes only, so the internal form
// is free.  monty.py's field.c form for this prime is 5 x 52 bits, Montgomery with R = 2^260 (monty.py:2129-2253); a
// 52-bit limb product costs four v_mad_u64_u32 and the pieces have to be re-packed into 52-bit limbs after every
// operation.  Here every partial product is ONE v_mad_i64_i32 into a 64-bit column, additions and subtractions are ten
// 32-bit adds with NO reduction (signed limbs: a - b needs no multiple of p, which for this prime -- three zero digits in
// the middle -- would have to be as large as 64p), and the reduction is interleaved column by column:
//     p = -1 + 2^18 B^3 + 2^10 B^7 + (B - 2^16) B^8 + (2^22 - 1) B^9,   B = 2^26,   p = -1 mod B  (so ndash = 1:
//     the Montgomery digit of a column IS its low 26 bits, and adding digit * (-1) clears them -- the same observation
//     monty.py:2239-2244 makes for 52-bit digits), the four other terms are multiply-adds into later columns.
// ELEVEN digits are reduced, not ten (R' = B^11 = 2^286): the product of two values |f|,|g| < 64p then comes out in
// (-2^-18 p, (1 + 2^-18) p), whatever lazy sums went in, so no operation ever needs a separate normalisation; the price
// is 4 of the 144 multiply-adds.  A field.c value x~ = x 2^260 enters with one multiplication by 2^312 mod p
// (x^ = x~ 2^26) and leaves through redc (x = x^ / R').
//
// Limb bounds: mul / sqr / mulc outputs have limbs 0..8 in [0, 2^26) and a small signed top limb ("K = 1").  A value
// built from such outputs by additions / subtractions has |limb| <= K 2^26 with K the number of terms; a product needs
// K_f K_g <= 190 (ten products per column below 2^63) and |limb| < 2^31.  wn26.h states K at every step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "field.h"

namespace ma {

struct Fm26 {
    static constexpr int32_t M26 = (1 << 26) - 1;
    static constexpr int32_t P8 = 0x3ff0000;     // digit 8 of p in the all-positive form: B - 2^16
    static constexpr int32_t P9 = 0x3fffff;      // digit 9: 2^22 - 1

    static constexpr int32_t prime(int i) {      // canonical digits of p
        return i < 3 ? M26 : i == 3 ? 0x3ffff : i < 7 ? 0 : i == 7 ? 0x400 : i == 8 ? P8 : P9;
    }
    static constexpr int32_t one(int i) {        // 2^286 mod p: the integer 1 in this form
        constexpr int32_t v[10] = {0x0, 0x10, 0x0, 0x0, 0x3c00000, 0x3ffffff, 0x3ffffff, 0x3ffffff, 0x3ffbfff, 0xfffff};
        return v[i];
    }
    static constexpr int32_t c312(int i) {       // 2^312 mod p: field.c form (R = 2^260) -> this form (R' = 2^286)
        constexpr int32_t v[10] = {0xffffff, 0x0, 0x10, 0x40000, 0x3ff0000, 0x3bfffff, 0x3ffffff, 0x3ff, 0x3feff00, 0x3fffff};
        return v[i];
    }

    // Multiplicands as values the compiler knows nothing about.  Where it can prove a limb non-negative it turns the sign
    // extension of the 64-bit product into a zero extension; a signed x zero-extended product has no single instruction
    // (v_mad_i64_i32 wants two sign extensions, v_mad_u64_u32 two zero extensions) and becomes two multiply-adds plus moves
    // (measured: +450 multiply-adds in the addition of the main loop).  No instruction is emitted for this.
    static MA_DEV void opaque(const int32_t* f, int32_t* r) {
        static_for<0, 10>([&](auto I) {
            int32_t x = f[I];
#if defined(__HIP_DEVICE_COMPILE__)
            asm("" : "+v"(x));
#endif
            r[I] = x;
        });
    }
    // MODE 0: r = f g / R';  MODE 1: r = f^2 / R';  MODE 2: r = f / R' (g unused);  MODE 3: r = (f g + u v) / R' -- two
    // products under ONE reduction (44 multiply-adds saved); needs K_f K_g + K_u K_v <= 190
    template <int MODE>
    static MA_DEV void mont(const int32_t* f, const int32_t* g, int32_t* r, const int32_t* u = nullptr, const int32_t* v = nullptr) {
        int32_t fo[10], go[10], uo[10], vo[10];
        opaque(f, fo);
        if constexpr (MODE == 0 || MODE == 3) opaque(g, go);
        if constexpr (MODE == 3) { opaque(u, uo); opaque(v, vo); }
        f = fo; g = go; u = uo; v = vo;
        int32_t f2[10];
        if constexpr (MODE == 1) static_for<0, 10>([&](auto I) { f2[I] = (int32_t)(2u * (uint32_t)f[I]); });
        int32_t m[11], t[10];
        int64_t c = 0;
        // 2^18 and 2^10 as opaque scalar registers: written as constants the compiler turns each of the 22 products into a
        // 64-bit shift plus a 64-bit add (v_lshl_add_u64 shifts by 0..4 only), two 5-cycle instructions instead of one
        int32_t c18 = 1 << 18, c10 = 1 << 10;
#if defined(__HIP_DEVICE_COMPILE__)
        asm("s_mov_b32 %0, 0x40000" : "=s"(c18));
        asm("s_mov_b32 %0, 0x400" : "=s"(c10));
#endif
        static_for<0, 20>([&](auto KK) {
            constexpr int k = KK;
            int64_t acc = c;
            if constexpr (MODE == 2) {
                if constexpr (k < 10) acc += f[k];
            } else {
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
            }
            // digit l of the Montgomery multiplier times the digits 3, 7, 8, 9 of p
            if constexpr (k - 3 >= 0 && k - 3 <= 10) { acc += (int64_t)m[k - 3] * c18; MA_PIN(acc); }
            if constexpr (k - 7 >= 0 && k - 7 <= 10) { acc += (int64_t)m[k - 7] * c10; MA_PIN(acc); }
            if constexpr (k - 8 >= 0 && k - 8 <= 10) { acc += (int64_t)m[k - 8] * P8; MA_PIN(acc); }
            if constexpr (k - 9 >= 0 && k - 9 <= 10) { acc += (int64_t)m[k - 9] * P9; MA_PIN(acc); }
            const int32_t lo = (int32_t)((uint32_t)acc & (uint32_t)M26);
            if constexpr (k <= 10) m[k] = lo;       // + digit * (-1) clears the low 26 bits: the column carries on exactly
            else t[k - 11] = lo;
            c = acc >> 26;
        });
        t[9] = (int32_t)c;
        static_for<0, 10>([&](auto I) { r[I] = t[I]; });
    }
    static MA_DEV void mul(const int32_t* f, const int32_t* g, int32_t* r) { mont<0>(f, g, r); }
    static MA_DEV void sqr(const int32_t* f, int32_t* r) { mont<1>(f, f, r); }
    static MA_DEV void redc(const int32_t* f, int32_t* r) { mont<2>(f, f, r); }
    static MA_DEV void mul2(const int32_t* f, const int32_t* g, const int32_t* u, const int32_t* v, int32_t* r) { mont<3>(f, g, r, u, v); }

    // limb-wise, in WRAPPING 32-bit arithmetic: with the signed operators (overflow undefined) the compiler is entitled to do
    // the addition in 64 bits after sign extension, and then multiplies the 64-bit sum with two v_mad_u64_u32 plus moves
    // instead of one v_mad_i64_i32 (measured: 1 959 instead of 1 422 multiply-adds in the secp256k1 addition)
    static MA_DEV void add(const int32_t* f, const int32_t* g, int32_t* r) { static_for<0, 10>([&](auto I) { r[I] = (int32_t)((uint32_t)f[I] + (uint32_t)g[I]); }); }
    static MA_DEV void sub(const int32_t* f, const int32_t* g, int32_t* r) { static_for<0, 10>([&](auto I) { r[I] = (int32_t)((uint32_t)f[I] - (uint32_t)g[I]); }); }
    static MA_DEV void neg(const int32_t* f, int32_t* r) { static_for<0, 10>([&](auto I) { r[I] = (int32_t)(0u - (uint32_t)f[I]); }); }
    static MA_DEV void copy(const int32_t* f, int32_t* r) { static_for<0, 10>([&](auto I) { r[I] = f[I]; }); }
    static MA_DEV void zero(int32_t* r) { static_for<0, 10>([&](auto I) { r[I] = 0; }); }
    static MA_DEV void set_one(int32_t* r) { static_for<0, 10>([&](auto I) { r[I] = one(I); }); }
    // r = s ? g : f per lane (v_cndmask; both values are read before the choice)
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

    // z^(p-2), p - 2 = ffffffff 00000001 00000000 00000000 00000000 ffffffff ffffffff fffffffd: 255 squarings, 12 multiplications
    static MA_DEV void invert(const int32_t* z, int32_t* out) {
        int32_t x2[10], x3[10], x6[10], x12[10], x15[10], x30[10], x32[10], t[10];
        sqr(z, x2);   mul(x2, z, x2);                        // 2^2 - 1
        sqr(x2, x3);  mul(x3, z, x3);                        // 2^3 - 1
        copy(x3, x6);   sqn(x6, 3);   mul(x6, x3, x6);       // 2^6 - 1
        copy(x6, x12);  sqn(x12, 6);  mul(x12, x6, x12);     // 2^12 - 1
        copy(x12, x15); sqn(x15, 3);  mul(x15, x3, x15);     // 2^15 - 1
        copy(x15, x30); sqn(x30, 15); mul(x30, x15, x30);    // 2^30 - 1
        copy(x30, x32); sqn(x32, 2);  mul(x32, x2, x32);     // 2^32 - 1
        copy(x32, t);
        sqn(t, 32);  mul(t, z, t);                           // ffffffff 00000001
        sqn(t, 128); mul(t, x32, t);                         // ... 00000000 00000000 00000000 ffffffff
        sqn(t, 32);  mul(t, x32, t);                         // ... ffffffff
        sqn(t, 30);  mul(t, x30, t);
        sqn(t, 2);   mul(t, z, out);                         // ... fffffffd
    }

    // field.c form (5 x 52-bit limbs of x 2^260 mod p, limbs below 2^54: the contract of the curve layer) -> this form
    static MA_DEV void from52(const spint* x, int32_t* r) {
        int32_t h[10], c[10];
        static_for<0, 5>([&](auto K) {
            constexpr int k = K;
            h[2 * k] = (int32_t)((uint32_t)x[k] & (uint32_t)M26);
            h[2 * k + 1] = (int32_t)(x[k] >> 26);               // < 2^28
        });
        static_for<0, 10>([&](auto I) { c[I] = c312(I); });
        mul(h, c, r);
    }
    // the integer value mod p, canonical, as four little-endian 64-bit words
    static MA_DEV void to_words(const int32_t* f, uint64_t* w) {
        int32_t t[10], s[10];
        redc(f, t);                                   // value in [0, p]: limbs 0..8 in [0, 2^26), 0 <= t[9] <= 2^22
        int32_t bw = 0;                               // s = t - p with borrow
        static_for<0, 10>([&](auto I) {
            constexpr int i = I;
            const int32_t d = t[i] - prime(i) + bw;
            bw = d >> 31;
            s[i] = (i < 9) ? (d & M26) : d;
        });
        const bool ge = bw == 0;                      // t >= p (only t == p can occur)
        static_for<0, 10>([&](auto I) { t[I] = ge ? s[I] : t[I]; });
        static_for<0, 4>([&](auto K) { w[K] = 0; });
        static_for<0, 10>([&](auto I) {
            constexpr int i = I;
            constexpr int o = 26 * i, wi = o / 64, sh = o % 64;
            w[wi] |= (uint64_t)(uint32_t)t[i] << sh;
            if constexpr (sh + 26 > 64 && wi + 1 < 4) w[wi + 1] |= (uint64_t)(uint32_t)t[i] >> (64 - sh);
        });
    }
    // two limbs per 64-bit word (the window tables are stored like this)
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
