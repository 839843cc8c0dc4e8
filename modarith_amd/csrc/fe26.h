// modarith_amd/csrc/fe26.h -- GF(2^255-19) in ten 25.5-bit limbs for the fused X25519 ladder on gfx950.
//
// Why a second representation: inside the fused ladder only the final 32 output bytes are compared
// with the reference (rfc7748.c:254 modexp does a full redc, so they are canonical), which frees the
// internal limb form (SURVEY 7 "hard parts", 8(f4)).  On CDNA4 the native wide multiplier is
// v_mad_u64_u32 (32x32+64 -> 64, measured ~2 issue slots); a 51-bit limb product costs four of them plus
// a 128-bit carry fix-up (about 500 VALU instructions per modmul in the bit-exact 5x51 form).  With
// limbs below 2^32 every partial product is ONE v_mad_u64_u32 accumulating straight into a 64-bit column
// register: 100 of them per multiplication, 55 per squaring, no cross-word carries.
//
// Radix 2^25.5: limb i carries 26 bits (i even) or 25 bits (i odd); value = sum f_i * 2^ceil(25.5 i).
// Unsigned limbs.  "tight" = as left by carry(): f_even < 2^26, f_odd < 2^25 (f_1 < 2^25 + 2^16).
// add()/sub() of tight operands give limbs < 1.5*2^27 (even) / 1.5*2^26 (odd); mul()/sqr() accept such
// operands on both sides: the largest column is 124.5 * (1.5*2^27)^2 < 2^62.2, and the pre-multiplied
// factors 19*g_j, 38*f_j, 2*f_i stay below 2^32.  The ladder never multiplies anything that is more
// than one add/sub away from a tight value (same discipline as the reference's generic=False form,
// pseudo.py:1523-1528).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "field.h"
#include "fe_finish.h"

namespace ma {

struct Fe26 {
    static constexpr uint32_t M26 = (1u << 26) - 1, M25 = (1u << 25) - 1;
    // bit position of limb i
    static constexpr int pos(int i) { return (51 * i + 1) / 2; }   // 0,26,51,77,102,128,153,179,204,230
    static constexpr int bits(int i) { return (i & 1) ? 25 : 26; }

    // reduce ten 64-bit columns to tight limbs
    static MA_DEV void carry(uint64_t* h, uint32_t* r) {
        static_for<0, 9>([&](auto I) {
            constexpr int i = I;
            h[i + 1] += h[i] >> bits(i);
            r[i] = (uint32_t)h[i] & ((i & 1) ? M25 : M26);
        });
        uint64_t c9 = h[9] >> 25;
        r[9] = (uint32_t)h[9] & M25;
        uint64_t h0 = (uint64_t)r[0] + 19 * c9;      // c9 < 2^38: fits
        r[0] = (uint32_t)h0 & M26;
        r[1] += (uint32_t)(h0 >> 26);
    }

    // Columns are computed in order and the carry out of column k is the INITIAL accumulator value of column
    // k+1, so it rides for free in that column's first v_mad_u64_u32 (no separate 64-bit add per limb).
    static MA_DEV void wrap(uint64_t c9, uint32_t* r) {
        uint64_t h0 = (uint64_t)r[0] + 19 * c9;      // c9 < 2^38
        r[0] = (uint32_t)h0 & M26;
        r[1] += (uint32_t)(h0 >> 26);
    }
    static MA_DEV void pre19(const uint32_t* g, uint32_t* g19) {
        static_for<1, 10>([&](auto J) { g19[J] = 19u * g[J]; });
    }
    static MA_DEV void mul(const uint32_t* f, const uint32_t* g, uint32_t* r) {
        uint32_t g19[10];
        pre19(g, g19);
        mul(f, g, g19, r);
    }
    // g19 = 19*g[1..9] supplied by the caller (the ladder's x1 is multiplied in every step: computed once)
    static MA_DEV void mul(const uint32_t* f, const uint32_t* g, const uint32_t* g19, uint32_t* r) {
        uint32_t f2[10];
        static_for<0, 5>([&](auto K) { f2[2 * K + 1] = 2u * f[2 * K + 1]; });
        uint64_t c = 0;
        uint32_t t[10];
        static_for<0, 10>([&](auto KK) {
            constexpr int k = KK;
            uint64_t acc = c;
            static_for<0, 10>([&](auto II) {
                constexpr int i = II;
                constexpr int j = (k - i + 10) % 10;
                constexpr bool wrp = (i + j) >= 10;
                constexpr bool dbl = (i & 1) && (j & 1);
                const uint32_t a = dbl ? f2[i] : f[i];
                const uint32_t b = wrp ? g19[j] : g[j];
                acc += (uint64_t)a * b;
                MA_PIN(acc);
            });
            t[k] = (uint32_t)acc & ((k & 1) ? M25 : M26);
            c = acc >> bits(k);
        });
        static_for<0, 10>([&](auto I) { r[I] = t[I]; });
        wrap(c, r);
    }

    static MA_DEV void sqr(const uint32_t* f, uint32_t* r) {
        uint32_t f2[10], f19[10], f38[10];
        static_for<0, 10>([&](auto I) { f2[I] = 2u * f[I]; });
        static_for<5, 10>([&](auto J) { f19[J] = 19u * f[J]; });
        static_for<0, 3>([&](auto K) { f38[2 * K + 5] = 38u * f[2 * K + 5]; });   // odd j >= 5
        uint64_t c = 0;
        uint32_t t[10];
        static_for<0, 10>([&](auto KK) {
            constexpr int k = KK;
            uint64_t acc = c;
            static_for<0, 10>([&](auto II) {
                constexpr int i = II;
                constexpr int j = (k - i + 10) % 10;
                if constexpr (i <= j) {
                    constexpr bool wrp = (i + j) >= 10;
                    constexpr bool odd2 = (i & 1) && (j & 1);
                    uint32_t a, b;
                    if constexpr (i == j) {
                        a = odd2 ? f2[i] : f[i];
                        b = wrp ? f19[j] : f[j];
                    } else {
                        a = f2[i];                               // symmetric term counted twice
                        b = wrp ? (odd2 ? f38[j] : f19[j]) : (odd2 ? f2[j] : f[j]);
                    }
                    acc += (uint64_t)a * b;
                    MA_PIN(acc);
                }
            });
            t[k] = (uint32_t)acc & ((k & 1) ? M25 : M26);
            c = acc >> bits(k);
        });
        static_for<0, 10>([&](auto I) { r[I] = t[I]; });
        wrap(c, r);
    }

    // r = f * c for a small constant (a24 = 121665)
    template <uint32_t C>
    static MA_DEV void mul_small(const uint32_t* f, uint32_t* r) {
        uint64_t h[10];
        static_for<0, 10>([&](auto I) { h[I] = (uint64_t)f[I] * C; });
        carry(h, r);
    }

    // r = f * C + a, carried like mul(): the carry out of limb i is the accumulator start of limb i+1 (rides in its
    // multiply-add), the addend joins the masked digit with one 32-bit add.  f < 1.5*2^27, C < 2^17: carries < 2^20;
    // a tight -> r < 2^27 (even) / 2^26 (odd), r[1] a few units more: one add away from tight, as mul()/sqr() accept.
    template <uint32_t C>
    static MA_DEV void mul_small_add(const uint32_t* f, const uint32_t* a, uint32_t* r) {
        uint64_t c = 0;
        uint32_t t[10];
        static_for<0, 10>([&](auto KK) {
            constexpr int k = KK;
            const uint64_t acc = c + (uint64_t)f[k] * C;
            MA_PIN(acc);
            t[k] = (uint32_t)acc & ((k & 1) ? M25 : M26);
            c = acc >> bits(k);
        });
        wrap(c, t);
        static_for<0, 10>([&](auto I) { r[I] = t[I] + a[I]; });
    }

    static MA_DEV void add(const uint32_t* f, const uint32_t* g, uint32_t* r) {
        static_for<0, 10>([&](auto I) { r[I] = f[I] + g[I]; });
    }
    // r = f - g + 2p  (limbs of 2p: 2^27-38, 2^26-2, 2^27-2, 2^26-2, ...)
    static MA_DEV void sub(const uint32_t* f, const uint32_t* g, uint32_t* r) {
        static_for<0, 10>([&](auto I) {
            constexpr int i = I;
            constexpr uint32_t twop = (i == 0) ? 0x7ffffdau : ((i & 1) ? 0x3fffffeu : 0x7fffffeu);
            r[i] = (f[i] + twop) - g[i];
        });
    }
    // constant-time swap by masking (mask = 0 or ~0 per lane); measured equal to v_cndmask pairs
    static MA_DEV void cswap(uint32_t mask, uint32_t* f, uint32_t* g) {
        static_for<0, 10>([&](auto I) {
            uint32_t t = (f[I] ^ g[I]) & mask;
            f[I] ^= t;
            g[I] ^= t;
        });
    }
    // r = s ? g : f per lane (v_cndmask; both values are read before the choice)
    static MA_DEV void select(bool s, const uint32_t* f, const uint32_t* g, uint32_t* r) {
        static_for<0, 10>([&](auto I) {
            const uint32_t x = f[I], y = g[I];
            r[I] = s ? y : x;
        });
    }
    static MA_DEV void copy(const uint32_t* f, uint32_t* r) { static_for<0, 10>([&](auto I) { r[I] = f[I]; }); }
    static MA_DEV void set(uint32_t v, uint32_t* r) { static_for<0, 10>([&](auto I) { r[I] = (I == 0) ? v : 0u; }); }

    static MA_DEV void sqn(uint32_t* f, int n) {
#pragma unroll 1
        for (int i = 0; i < n; i++) sqr(f, f);
    }

    // z^(p-2) = z^(2^255-21): 254 squarings + 11 multiplications (run ladder 1,2,4,5,10,20,40,50,100,200,250)
    static MA_DEV void invert(const uint32_t* z, uint32_t* out) {
        uint32_t t0[10], t1[10], t2[10], t3[10];
        sqr(z, t0);                                   // 2
        sqr(t0, t1); sqr(t1, t1);                     // 8
        mul(z, t1, t1);                               // 9
        mul(t0, t1, t0);                              // 11
        sqr(t0, t2);                                  // 22
        mul(t1, t2, t1);                              // 31 = 2^5-1
        copy(t1, t2); sqn(t2, 5);  mul(t2, t1, t1);   // 2^10-1
        copy(t1, t2); sqn(t2, 10); mul(t2, t1, t2);   // 2^20-1
        copy(t2, t3); sqn(t3, 20); mul(t3, t2, t2);   // 2^40-1
        sqn(t2, 10);               mul(t2, t1, t1);   // 2^50-1
        copy(t1, t2); sqn(t2, 50); mul(t2, t1, t2);   // 2^100-1
        copy(t2, t3); sqn(t3, 100); mul(t3, t2, t2);  // 2^200-1
        sqn(t2, 50);               mul(t2, t1, t1);   // 2^250-1
        sqn(t1, 5);                mul(t1, t0, out);  // 2^255-21
    }

    // 255-bit little-endian integer in four 64-bit words (bit 255 already cleared) -> limbs (tight).
    // Non-canonical inputs (>= p) are fine: the value is reduced at export, as modimp/modfsb would.
    static MA_DEV void from_words(const uint64_t* w, uint32_t* r) {
        static_for<0, 10>([&](auto I) {
            constexpr int i = I;
            constexpr int o = pos(i), wi = o / 64, sh = o % 64;
            uint64_t v = w[wi] >> sh;
            if constexpr (sh + bits(i) > 64 && wi + 1 < 4) v |= w[wi + 1] << (64 - sh);
            r[i] = (uint32_t)v & ((i & 1) ? M25 : M26);
        });
    }
    // canonical export: value mod p as four little-endian 64-bit words
    static MA_DEV void to_words(const uint32_t* f, uint64_t* w) {
        uint64_t h[10];
        uint32_t t[10];
        static_for<0, 10>([&](auto I) { h[I] = f[I]; });
        carry(h, t);                                   // tight (t1 may carry 2^16 extra)
        static_for<0, 10>([&](auto I) { h[I] = t[I]; });
        carry(h, t);                                   // now every limb strictly within its width, value < 2^255 + small
        // q = 1 iff t >= p  <=>  t + 19 >= 2^255
        uint32_t q = (t[0] + 19u) >> 26;
        static_for<1, 10>([&](auto I) {
            constexpr int i = I;
            q = (t[i] + q) >> bits(i);
        });
        uint32_t c = 19u * q;                          // add 19 if >= p, then drop bit 255
        static_for<0, 10>([&](auto I) {
            constexpr int i = I;
            uint32_t s = t[i] + c;
            c = s >> bits(i);
            t[i] = s & ((i & 1) ? M25 : M26);
        });
        static_for<0, 4>([&](auto K) { w[K] = 0; });
        static_for<0, 10>([&](auto I) {
            constexpr int i = I;
            constexpr int o = pos(i), wi = o / 64, sh = o % 64;
            w[wi] |= (uint64_t)t[i] << sh;
            if constexpr (sh + bits(i) > 64 && wi + 1 < 4) w[wi + 1] |= (uint64_t)t[i] >> (64 - sh);
        });
    }
};

// One X25519 scalar multiplication (rfc7748.c:156-256) on the fe26 representation: kw, uw = the 32-byte scalar and
// u-coordinate records as four little-endian words; ow = the canonical result.
//
// Ladder step without data movement for the conditional swap.  With A = x2+z2, B = x2-z2, C = x3+z3, D = x3-z3 taken
// from the UNSWAPPED pairs, exchanging (x2,z2) <-> (x3,z3) exchanges A <-> C and B <-> D, hence DA <-> CB: the
// differential addition x3' = (DA+CB)^2, z3' = x1 (DA-CB)^2 does not see the swap at all (the difference only changes
// sign under the square), and the doubling only needs A' = swap ? C : A and B' = swap ? D : B.  Two 10-limb selects
// replace the two 10-limb swaps of rfc7748.c:190-191 (20 v_cndmask instead of 40 + mask arithmetic); the state after
// every step is exactly the reference's ("2" = the doubled point, "3" = the sum), so the swap bit chains as there.
// the ladder proper: leaves (x2 : z2) of k*u, tight limbs
MA_DEV void x25519_fe26_ladder(const uint64_t* kw_in, const uint64_t* uw_in, uint32_t* x2, uint32_t* z2) {
    using F = Fe26;
    uint64_t kw[4], uw[4];
    static_for<0, 4>([&](auto K) { kw[K] = kw_in[K]; uw[K] = uw_in[K]; });
    uw[3] &= 0x7fffffffffffffffull;                     // mask bit 255 of u (rfc7748.c:171-172)
    kw[0] &= ~7ull;                                     // clamp (rfc7748.c:135-141)
    kw[3] = (kw[3] & 0x7fffffffffffffffull) | 0x4000000000000000ull;
    // left-align: bit 254 -> bit 63 of kw[3]
    kw[3] = (kw[3] << 1) | (kw[2] >> 63);
    kw[2] = (kw[2] << 1) | (kw[1] >> 63);
    kw[1] = (kw[1] << 1) | (kw[0] >> 63);
    kw[0] <<= 1;

    uint32_t x1[10], x1_19[10], x3[10], z3[10];
    F::from_words(uw, x1);
    F::pre19(x1, x1_19);
    x1_19[0] = 0;
    F::set(1, x2);
    F::set(0, z2);
    F::copy(x1, x3);
    F::set(1, z3);

    uint32_t swap = 0;
#pragma unroll 1
    for (int step = 0; step < 255; step++) {
        const uint32_t kt = (uint32_t)(kw[3] >> 63);
        kw[3] = (kw[3] << 1) | (kw[2] >> 63);
        kw[2] = (kw[2] << 1) | (kw[1] >> 63);
        kw[1] = (kw[1] << 1) | (kw[0] >> 63);
        kw[0] <<= 1;
        const bool sw = (swap ^ kt) != 0;
        swap = kt;
        uint32_t A[10], B[10], C[10], D[10], As[10], Bs[10], AA[10], BB[10], E[10];
        F::add(x2, z2, A);
        F::add(x3, z3, C);
        F::sub(x2, z2, B);
        F::sub(x3, z3, D);
        F::select(sw, A, C, As);
        F::select(sw, B, D, Bs);
        F::mul(D, A, D);          // DA  (CB when swapped: the pair {DA, CB} is swap-invariant)
        F::mul(C, B, C);          // CB
        F::sqr(As, AA);
        F::sqr(Bs, BB);
        F::sub(D, C, z3);
        F::add(D, C, x3);
        F::sub(AA, BB, E);
        F::mul_small_add<121665>(E, AA, z2);      // AA + a24*E
        F::mul(z2, E, z2);
        F::sqr(x3, x3);
        F::sqr(z3, z3);
        F::mul(z3, x1, x1_19, z3);
        F::mul(AA, BB, x2);
    }
    F::select(swap != 0, x2, x3, x2);
    F::select(swap != 0, z2, z3, z2);
}

MA_DEV void x25519_fe26_one(const uint64_t* kw_in, const uint64_t* uw_in, uint64_t* ow) {
    using F = Fe26;
    uint32_t x2[10], z2[10];
    x25519_fe26_ladder(kw_in, uw_in, x2, z2);
    F::invert(z2, z2);
    F::mul(x2, z2, x2);
    F::to_words(x2, ow);
}

#ifdef MA_LADDER_FE26   // the kernel is emitted by the unit that owns the ladder entry point (capi_prime.inc)
// Batched X25519 on the fe26 representation: contiguous 32-byte records (simd/rfc7748_simt.cu:165-168), one per lane.
__global__ __launch_bounds__(256) void k_x25519_fe26(const uint64_t* bk, const uint64_t* bu, uint64_t* bv, size_t n) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        uint64_t kw[4], uw[4], ow[4];
        static_for<0, 4>([&](auto K) { kw[K] = bk[t * 4 + K]; });
        static_for<0, 4>([&](auto K) { uw[K] = bu[t * 4 + K]; });
        x25519_fe26_one(kw, uw, ow);
        static_for<0, 4>([&](auto K) { bv[t * 4 + K] = ow[K]; });
    }
}

// the split form: ladders only; canonical x2 -> the output record, canonical z2 -> wz[4][n] (word-major)
__global__ __launch_bounds__(256) void k_x25519_fe26_xz(const uint64_t* bk, const uint64_t* bu, uint64_t* bv, uint64_t* wz, size_t n) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        uint64_t kw[4], uw[4], xw[4], zw[4];
        static_for<0, 4>([&](auto K) { kw[K] = bk[t * 4 + K]; });
        static_for<0, 4>([&](auto K) { uw[K] = bu[t * 4 + K]; });
        uint32_t x2[10], z2[10];
        x25519_fe26_ladder(kw, uw, x2, z2);
        Fe26::to_words(x2, xw);
        Fe26::to_words(z2, zw);
        static_for<0, 4>([&](auto K) { bv[t * 4 + K] = xw[K]; });
        static_for<0, 4>([&](auto K) { wz[(size_t)K * n + t] = zw[K]; });
    }
}
#endif

}  // namespace ma
