// modarith_amd/csrc/fe28.h -- GF(2^448 - 2^224 - 1) in sixteen 28-bit limbs for the fused X448 ladder on gfx950.
//
// Same idea as fe26.h: rfc7748() ends in modexp (full redc), so only canonical bytes leave the kernel and the
// internal limb form is free.  With limbs below 2^32 every partial product is one v_mad_u64_u32 into a 64-bit
// column register (256 per multiplication, 136 per squaring) instead of four plus a 128-bit fix-up for the
// 56-bit limbs of the bit-exact field.
//
// Radix 2^28, value = sum f_i 2^(28 i), phi = 2^224 is limb 8, 2^448 = phi + 1 (mod p).
// "tight" = as left by carry(): f_i < 2^28 (f_1, f_9 < 2^28 + 2^9).  add() of tight operands gives limbs < 2^29;
// sub() adds 4p and carries back to tight.  mul()/sqr() accept limbs < 2^29 on both sides: the raw columns
// c_0..c_30 hold at most 16 products of 2^58, and the folded column r_m = c_m + c_{m+8} + 2 c_{m+16} (m >= 8) or
// c_m + c_{m+16} + c_{m+24} (m < 8) at most 38 of them: < 2^63.3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "field.h"
#include "fe_finish.h"

namespace ma {

struct Fe28 {
    static constexpr uint32_t M28 = (1u << 28) - 1;

    // fold the 31 raw columns (2^448 = 2^224 + 1) and carry to tight limbs
    static MA_DEV void reduce(const uint64_t* c, uint32_t* r) {
        uint64_t h[16];
        static_for<0, 16>([&](auto MM) {
            constexpr int m = MM;
            uint64_t v = c[m];
            if constexpr (m < 8) {
                v += c[m + 16];
                if constexpr (m + 24 <= 30) v += c[m + 24];
            } else {
                v += c[m + 8];
                if constexpr (m + 16 <= 30) v += 2 * c[m + 16];
            }
            h[m] = v;
        });
        carry(h, r);
    }

    // sixteen 64-bit columns -> tight limbs; the carry out of limb 15 re-enters at limbs 0 and 8
    static MA_DEV void carry(uint64_t* h, uint32_t* r) {
        static_for<0, 15>([&](auto I) {
            constexpr int i = I;
            h[i + 1] += h[i] >> 28;
            r[i] = (uint32_t)h[i] & M28;
        });
        const uint64_t top = h[15] >> 28;          // < 2^36
        r[15] = (uint32_t)h[15] & M28;
        const uint64_t h0 = (uint64_t)r[0] + top;
        const uint64_t h8 = (uint64_t)r[8] + top;
        r[0] = (uint32_t)h0 & M28;
        r[1] += (uint32_t)(h0 >> 28);
        r[8] = (uint32_t)h8 & M28;
        r[9] += (uint32_t)(h8 >> 28);
    }

    static MA_DEV void mul(const uint32_t* f, const uint32_t* g, uint32_t* r) {
        uint64_t c[31];
        static_for<0, 31>([&](auto KK) {
            constexpr int k = KK;
            constexpr int lo = k < 16 ? 0 : k - 15, hi = k < 16 ? k : 15;
            uint64_t acc = 0;
            static_for<lo, hi + 1>([&](auto II) {
                constexpr int i = II;
                acc += (uint64_t)f[i] * g[k - i];
            });
            c[k] = acc;
        });
        reduce(c, r);
    }

    static MA_DEV void sqr(const uint32_t* f, uint32_t* r) {
        uint32_t f2[16];
        static_for<0, 16>([&](auto I) { f2[I] = 2u * f[I]; });
        uint64_t c[31];
        static_for<0, 31>([&](auto KK) {
            constexpr int k = KK;
            constexpr int lo = k < 16 ? 0 : k - 15, hi = k < 16 ? k : 15;
            uint64_t acc = 0;
            static_for<lo, hi + 1>([&](auto II) {
                constexpr int i = II;
                constexpr int j = k - i;
                if constexpr (i < j) acc += (uint64_t)f2[i] * f[j];
                else if constexpr (i == j) acc += (uint64_t)f[i] * f[i];
            });
            c[k] = acc;
        });
        reduce(c, r);
    }

    // Karatsuba on the golden-ratio structure: with a = a0 + a1*phi, phi = 2^224, phi^2 = phi + 1,
    //   a*b = (a0 b0 + a1 b1) + ((a0 + a1)(b0 + b1) - a0 b0) * phi
    // three 8x8 products (192 v_mad_u64_u32 instead of 256).  Needs one operand tight and the other below 2^29 (or
    // both tight for the squaring): the half sums are then below 2^29+2^10 and 2^30, a column of the middle product
    // holds 8 products below 2^59 (< 2^62), and a folded limb L_m + H_{m-8} + H_m stays below 2^63.4.
    static MA_DEV void fold_lh(const uint64_t* L, const uint64_t* H, uint32_t* r) {
        uint64_t h[16];
        static_for<0, 16>([&](auto MM) {
            constexpr int m = MM;
            uint64_t v = 0;
            if constexpr (m <= 14) v = L[m];
            if constexpr (m < 8) {
                if constexpr (m + 8 <= 14) v += H[m + 8];
            } else {
                v += H[m - 8];
                if constexpr (m <= 14) v += H[m];
            }
            h[m] = v;
        });
        carry(h, r);
    }
    // Column form of the Karatsuba product: with Z0 = a0 b0, Z2 = a1 b1, Z1 = (a0+a1)(b0+b1) (8 x 8 limbs each, 15 columns),
    //   h[m] = Z0[m] + Z2[m] + Z1[m+8] - Z0[m+8]          (m <= 6)       h[7]  = Z0[7] + Z2[7]
    //   h[m] = Z1[m-8] - Z0[m-8] + Z2[m] + Z1[m]           (8 <= m <= 14) h[15] = Z1[7] - Z0[7]
    // (Z0[m] cancels for m >= 8).  Only Z0[0..7] and Z1[8..14] occur twice; they are summed once (64 multiply-adds) and
    // added with 64-bit adds.  Every other product goes straight into the accumulator of its column, which starts from
    // the carry of the previous column (MA_PIN keeps that order), Z0[m+8] through a signed multiply-add with -a0: about
    // 38 64-bit add/sub instructions per product instead of 83 for "sum three 15-column products, fold, carry".
    // Same 192 multiply-adds, same bounds (the column values are the ones of the fold); the running value may be
    // negative in between (two's complement), each finished column is >= 0 because Z1 >= Z0 + Z2 column by column.
    template <bool SQR>
    static MA_DEV void karatsuba(const uint32_t* f, const uint32_t* g, uint32_t* r) {
        uint32_t fs[8], gs[8], nf[8];
        static_for<0, 8>([&](auto I) { fs[I] = f[I] + f[I + 8]; nf[I] = 0u - f[I]; });
        if constexpr (!SQR) static_for<0, 8>([&](auto I) { gs[I] = g[I] + g[I + 8]; });
        const uint32_t* gg = SQR ? f : g;
        const uint32_t* gss = SQR ? fs : gs;
        // one column of an 8 x 8 product a*b (b = a for the squaring: symmetric terms once, doubled operand)
        auto col = [&](auto KK, const uint32_t* a, const uint32_t* b, uint64_t acc, bool neg_a) -> uint64_t {
            constexpr int k = KK;
            constexpr int lo = k < 8 ? 0 : k - 7, hi = k < 8 ? k : 7;
            static_for<lo, hi + 1>([&](auto II) {
                constexpr int i = II, j = k - i;
                if constexpr (!SQR) {
                    if (neg_a) acc += (uint64_t)((int64_t)(int32_t)a[i] * (int64_t)(int32_t)b[j]);
                    else acc += (uint64_t)a[i] * b[j];
                    MA_PIN(acc);
                } else if constexpr (i <= j) {
                    const uint32_t bj = (i < j) ? 2u * b[j] : b[j];
                    if (neg_a) acc += (uint64_t)((int64_t)(int32_t)a[i] * (int64_t)(int32_t)bj);
                    else acc += (uint64_t)a[i] * bj;
                    MA_PIN(acc);
                }
            });
            return acc;
        };
        uint64_t S0[8], S1[15];
        static_for<0, 8>([&](auto K) { S0[K] = col(K, f, gg, 0, false); });
        static_for<8, 15>([&](auto K) { S1[K] = col(K, fs, gss, 0, false); });
        uint64_t cy = 0;
        uint32_t t[16];
        static_for<0, 16>([&](auto MM) {
            constexpr int m = MM;
            uint64_t acc = cy;
            if constexpr (m <= 7) {
                acc = col(std::integral_constant<int, m>{}, f + 8, gg + 8, acc, false);                    // Z2[m]
                if constexpr (m <= 6) {
                    acc = col(std::integral_constant<int, m + 8>{}, nf, gg, acc, true);                    // -Z0[m+8]
                    acc += S1[m + 8];
                }
                acc += S0[m];
            } else {
                acc = col(std::integral_constant<int, m - 8>{}, fs, gss, acc, false);                      // Z1[m-8]
                if constexpr (m <= 14) {
                    acc = col(std::integral_constant<int, m>{}, f + 8, gg + 8, acc, false);                // Z2[m]
                    acc += S1[m];
                }
                acc -= S0[m - 8];
            }
            t[m] = (uint32_t)acc & M28;
            cy = acc >> 28;
        });
        // the carry out of limb 15 re-enters at limbs 0 and 8 (2^448 = 2^224 + 1)
        const uint64_t h0 = (uint64_t)t[0] + cy, h8 = (uint64_t)t[8] + cy;
        t[0] = (uint32_t)h0 & M28;
        t[1] += (uint32_t)(h0 >> 28);
        t[8] = (uint32_t)h8 & M28;
        t[9] += (uint32_t)(h8 >> 28);
        static_for<0, 16>([&](auto I) { r[I] = t[I]; });
    }
    static MA_DEV void mul_k(const uint32_t* f, const uint32_t* g, uint32_t* r) { karatsuba<false>(f, g, r); }
    static MA_DEV void sqr_k(const uint32_t* f, uint32_t* r) { karatsuba<true>(f, f, r); }     // f tight

    template <uint32_t C>
    static MA_DEV void mul_small(const uint32_t* f, uint32_t* r) {
        uint64_t h[16];
        static_for<0, 16>([&](auto I) { h[I] = (uint64_t)f[I] * C; });
        carry(h, r);
    }
    // r = f * C + a with the carry of limb i riding in the multiply-add of limb i+1 (f tight, C < 2^16: carries < 2^16;
    // a tight -> r < 2^29, r[1] / r[9] a few units more)
    template <uint32_t C>
    static MA_DEV void mul_small_add(const uint32_t* f, const uint32_t* a, uint32_t* r) {
        uint64_t c = 0;
        uint32_t t[16];
        static_for<0, 16>([&](auto KK) {
            constexpr int k = KK;
            const uint64_t acc = c + (uint64_t)f[k] * C;
            MA_PIN(acc);
            t[k] = (uint32_t)acc & M28;
            c = acc >> 28;
        });
        const uint32_t top = (uint32_t)c;              // 2^448 = 2^224 + 1: re-enters at limbs 0 and 8
        const uint32_t h0 = t[0] + top, h8 = t[8] + top;
        t[0] = h0 & M28;
        t[1] += h0 >> 28;
        t[8] = h8 & M28;
        t[9] += h8 >> 28;
        static_for<0, 16>([&](auto I) { r[I] = t[I] + a[I]; });
    }

    static MA_DEV void add(const uint32_t* f, const uint32_t* g, uint32_t* r) {
        static_for<0, 16>([&](auto I) { r[I] = f[I] + g[I]; });
    }
    // r = f - g + 4p, carried back to tight (limbs of 4p: 2^30-4, limb 8: 2^30-8)
    // (f, g < 2^29 + 2^10, i.e. sums of two tight values: no limb of f + 4p - g goes below 0 and every one is below 2^31, so the
    // carries run in 32-bit registers.  Rounds 1-3 added 2p, whose limbs 2^29-2 do NOT cover a subtrahend that is itself a sum
    // when its limb is at the top of its range and the minuend's limb is 0 or 1 -- which no random input produces, and which the
    // doubling of the order-4 point (0, 1, p-1) does: found in round 4 when the table builder began to stash canonical values.)
    static MA_DEV void sub(const uint32_t* f, const uint32_t* g, uint32_t* r) {
        uint32_t h[16];
        static_for<0, 16>([&](auto I) {
            constexpr int i = I;
            constexpr uint32_t fourp = (i == 8) ? 0x3ffffff8u : 0x3ffffffcu;
            h[i] = (f[i] + fourp) - g[i];
        });
        static_for<0, 15>([&](auto I) {
            constexpr int i = I;
            h[i + 1] += h[i] >> 28;
            r[i] = h[i] & M28;
        });
        const uint32_t top = h[15] >> 28;
        r[15] = h[15] & M28;
        const uint32_t h0 = r[0] + top, h8 = r[8] + top;
        r[0] = h0 & M28;
        r[1] += h0 >> 28;
        r[8] = h8 & M28;
        r[9] += h8 >> 28;
    }
    // r = s ? g : f per lane (v_cndmask; both values are read before the choice)
    static MA_DEV void select(bool s, const uint32_t* f, const uint32_t* g, uint32_t* r) {
        static_for<0, 16>([&](auto I) {
            const uint32_t x = f[I], y = g[I];
            r[I] = s ? y : x;
        });
    }
    static MA_DEV void cswap(uint32_t mask, uint32_t* f, uint32_t* g) {
        static_for<0, 16>([&](auto I) {
            uint32_t t = (f[I] ^ g[I]) & mask;
            f[I] ^= t;
            g[I] ^= t;
        });
    }
    static MA_DEV void copy(const uint32_t* f, uint32_t* r) { static_for<0, 16>([&](auto I) { r[I] = f[I]; }); }
    static MA_DEV void set(uint32_t v, uint32_t* r) { static_for<0, 16>([&](auto I) { r[I] = (I == 0) ? v : 0u; }); }
    static MA_DEV void sqn(uint32_t* f, int n) {                    // f tight (results of mul / sqr)
#pragma unroll 1
        for (int i = 0; i < n; i++) sqr_k(f, f);
    }

    // z^(p-2), p-2 = 2^448 - 2^224 - 3 = [223 ones][0][222 ones][0][1]: 447 squarings + 13 multiplications
    static MA_DEV void invert(const uint32_t* z, uint32_t* out) {
        uint32_t a[16], b[16], d[16];
        sqr(z, a);   mul(a, z, a);                    // 2^2-1
        sqr(a, b);   mul(b, z, b);                    // 2^3-1
        copy(b, a);  sqn(a, 3);   mul(a, b, a);       // 2^6-1
        copy(a, d);  sqn(d, 6);   mul(d, a, d);       // 2^12-1
        copy(d, a);  sqn(a, 12);  mul(a, d, a);       // 2^24-1
        sqn(a, 3);                mul(a, b, a);       // 2^27-1
        copy(a, d);  sqn(d, 27);  mul(d, a, d);       // 2^54-1
        copy(d, a);  sqn(a, 54);  mul(a, d, a);       // 2^108-1
        sqn(a, 3);                mul(a, b, a);       // 2^111-1
        copy(a, d);  sqn(d, 111); mul(d, a, d);       // 2^222-1   (d)
        sqr(d, a);                mul(a, z, a);       // 2^223-1   (a)
        sqn(a, 223);              mul(a, d, a);       // [223 ones][0][222 ones]
        sqn(a, 2);                mul(a, z, out);     // ... [0][1]
    }

    // 448-bit little-endian integer in seven 64-bit words -> limbs (tight)
    static MA_DEV void from_words(const uint64_t* w, uint32_t* r) {
        static_for<0, 16>([&](auto I) {
            constexpr int i = I;
            constexpr int o = 28 * i, wi = o / 64, sh = o % 64;
            uint64_t v = w[wi] >> sh;
            if constexpr (sh + 28 > 64 && wi + 1 < 7) v |= w[wi + 1] << (64 - sh);
            r[i] = (uint32_t)v & M28;
        });
    }
    // canonical export: value mod p as seven little-endian 64-bit words
    static MA_DEV void to_words(const uint32_t* f, uint64_t* w) {
        uint64_t h[16];
        uint32_t t[16], u[16];
        static_for<0, 16>([&](auto I) { h[I] = f[I]; });
        carry(h, t);
        static_for<0, 16>([&](auto I) { h[I] = t[I]; });
        carry(h, t);                                   // every limb < 2^28 except t1/t9 by at most 1; value < 2^448 + small
        // u = t + 2^224 + 1; q = carry out of bit 448  <=>  t >= p
        uint32_t c = 1;
        static_for<0, 16>([&](auto I) {
            constexpr int i = I;
            uint32_t s = t[i] + c + ((i == 8) ? 1u : 0u);
            u[i] = s & M28;
            c = s >> 28;
        });
        const uint32_t mask = 0u - c;                  // all ones if t >= p: take u (= t - p), else keep t
        // t may still hold a limb equal to 2^28 (t1 / t9): normalise it with one more pass before selecting
        uint32_t cc = 0;
        static_for<0, 16>([&](auto I) {
            constexpr int i = I;
            uint32_t s = t[i] + cc;
            t[i] = s & M28;
            cc = s >> 28;
        });
        static_for<0, 16>([&](auto I) { t[I] = (u[I] & mask) | (t[I] & ~mask); });
        static_for<0, 7>([&](auto K) { w[K] = 0; });
        static_for<0, 16>([&](auto I) {
            constexpr int i = I;
            constexpr int o = 28 * i, wi = o / 64, sh = o % 64;
            w[wi] |= (uint64_t)t[i] << sh;
            if constexpr (sh + 28 > 64 && wi + 1 < 7) w[wi + 1] |= (uint64_t)t[i] >> (64 - sh);
        });
    }
};

// One X448 scalar multiplication (rfc7748.c:156-256, A24 = 39081, COF = 2) on the fe28 representation; kw, uw = the
// 56-byte records as seven little-endian words.  The conditional swap is folded into two selects exactly as in
// fe26.h (x25519_fe26_one): {DA, CB} is invariant under the swap, only the doubling needs A' and B'.
// the ladder proper: leaves (x2 : z2) of k*u, tight limbs
MA_DEV void x448_fe28_ladder(const uint64_t* kw_in, const uint64_t* uw_in, uint32_t* x2, uint32_t* z2) {
    using F = Fe28;
    uint64_t kw[7], uw[7];
    static_for<0, 7>([&](auto K) { kw[K] = kw_in[K]; uw[K] = uw_in[K]; });
    kw[0] &= ~3ull;                                       // clamp (rfc7748.c:135-141): Nbits % 8 == 0
    kw[6] |= 0x8000000000000000ull;                       // bit 447 set; already left-aligned

    uint32_t x1[16], x3[16], z3[16];
    F::from_words(uw, x1);
    F::set(1, x2);
    F::set(0, z2);
    F::copy(x1, x3);
    F::set(1, z3);

    uint32_t swap = 0;
#pragma unroll 1
    for (int step = 0; step < 448; step++) {
        const uint32_t kt = (uint32_t)(kw[6] >> 63);
        static_for<0, 7>([&](auto KK) {
            constexpr int k = 6 - KK;
            kw[k] <<= 1;
            if constexpr (k > 0) kw[k] |= kw[k - 1] >> 63;
        });
        const bool sw = (swap ^ kt) != 0;
        swap = kt;
        uint32_t A[16], B[16], C[16], D[16], As[16], Bs[16], AA[16], BB[16], E[16];
        F::add(x2, z2, A);
        F::add(x3, z3, C);
        F::sub(x2, z2, B);
        F::sub(x3, z3, D);
        F::select(sw, A, C, As);
        F::select(sw, B, D, Bs);
        F::mul_k(D, A, D);                  // D, B tight; A, C below 2^29
        F::mul_k(C, B, C);
        F::sqr(As, AA);                     // a sum of two tight values is not tight
        F::sqr_k(Bs, BB);
        F::sub(D, C, z3);
        F::add(D, C, x3);
        F::sub(AA, BB, E);
        F::mul_small_add<39081>(E, AA, z2); // AA + a24*E
        F::mul_k(z2, E, z2);
        F::sqr(x3, x3);                     // x3 = D + C is not tight
        F::sqr_k(z3, z3);
        F::mul_k(z3, x1, z3);
        F::mul_k(AA, BB, x2);
    }
    F::select(swap != 0, x2, x3, x2);
    F::select(swap != 0, z2, z3, z2);
}

MA_DEV void x448_fe28_one(const uint64_t* kw_in, const uint64_t* uw_in, uint64_t* ow) {
    using F = Fe28;
    uint32_t x2[16], z2[16];
    x448_fe28_ladder(kw_in, uw_in, x2, z2);
    F::invert(z2, z2);
    F::mul(x2, z2, x2);
    F::to_words(x2, ow);
}

#ifdef MA_LADDER_FE28   // the kernel is emitted by the unit that owns the ladder entry point (capi_prime.inc)
// Batched X448 on the fe28 representation: contiguous 56-byte records, one per lane.
__global__ __launch_bounds__(256, 2) void k_x448_fe28(const uint64_t* bk, const uint64_t* bu, uint64_t* bv, size_t n) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        uint64_t kw[7], uw[7], ow[7];
        static_for<0, 7>([&](auto K) { kw[K] = bk[t * 7 + K]; });
        static_for<0, 7>([&](auto K) { uw[K] = bu[t * 7 + K]; });
        x448_fe28_one(kw, uw, ow);
        static_for<0, 7>([&](auto K) { bv[t * 7 + K] = ow[K]; });
    }
}

// the split form (fe_finish.h): ladders only; canonical x2 -> the output record, canonical z2 -> wz[7][n] (word-major)
__global__ __launch_bounds__(256, 2) void k_x448_fe28_xz(const uint64_t* bk, const uint64_t* bu, uint64_t* bv, uint64_t* wz, size_t n) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        uint64_t kw[7], uw[7], xw[7], zw[7];
        static_for<0, 7>([&](auto K) { kw[K] = bk[t * 7 + K]; });
        static_for<0, 7>([&](auto K) { uw[K] = bu[t * 7 + K]; });
        uint32_t x2[16], z2[16];
        x448_fe28_ladder(kw, uw, x2, z2);
        Fe28::to_words(x2, xw);
        Fe28::to_words(z2, zw);
        static_for<0, 7>([&](auto K) { bv[t * 7 + K] = xw[K]; });
        static_for<0, 7>([&](auto K) { wz[(size_t)K * n + t] = zw[K]; });
    }
}

#endif

}  // namespace ma
