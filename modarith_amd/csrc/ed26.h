// modarith_amd/csrc/ed26.h -- fused ecnXXXmul + ecnXXXget for ED25519 on the fe26 representation (gfx950).
//
// The reference's callers end a scalar multiplication in ecnXXXget (ed448.c:182-184, nist256.c:155-161 pattern):
// affine coordinates as canonical big-endian bytes (modexp does a full redc, pseudo.py:1115-1127).  Only those bytes
// leave this kernel, so -- exactly as for rfc7748() in fe26.h -- the internal limb form AND the group-law formulas are
// free: ten 25.5-bit limbs (every partial product one v_mad_u64_u32), extended twisted-Edwards coordinates
// (X:Y:Z:T), a = -1, Hisil-Wong-Carter-Dawson doubling (4S + 3M, + 1M when T is needed) and mixed addition with a
// cached affine operand (y+x, y-x, 2dxy; 6M).  These formulas are COMPLETE on -x^2 + y^2 = 1 + d x^2 y^2 (a = -1 is a
// square and d a non-square mod 2^255-19): no exceptional cases for any pair of curve points, including the neutral
// element and the points of small order, so the result is the affine point the reference's edwards.c:73-145 formulas
// reach, for every input point ON the curve.  (For off-curve input neither side means anything; they may differ.)
//
// Scalar multiplication is fixed-window like edwards.c:435-482 -- every scalar takes the same instruction and address
// sequence -- with 3-bit signed digits so that the whole table {1,2,3,4}P fits the register file: 12 field elements
// held canonical and packed (4 x 64 bits each, 96 VGPRs); no workspace, no LDS, no table traffic.  A lookup scans
// all four entries with lane-predicated selects (v_cndmask), the digit's sign swaps y+x / y-x and negates 2dxy.
// Recoding without carries: e' = e + sum_{i<86} 4*8^i < 8^86, digit_i = window_i(e') - 4 in [-4, 3].
// Work per scalar: 255 doublings + 86 additions + table (2 doublings, 1 general addition, one shared inversion)
// + final inversion: about 2.37e5 multiply-adds, against 2.04e5 for the X25519 ladder.
//
// Limb bounds ("scale" 1 = 2^27 on even limbs / 2^26 on odd ones = twice tight): Fe26::mul(f, g) needs the
// 19-premultiplied operand g below scale 1.68 (19 g < 2^32) and scale(f) * scale(g) <= 8 (column < 2^64); sqr needs
// scale <= 1.68.  The comments give the scale of every intermediate; one weak carry per doubling keeps them there.
#pragma once
#include "fe26.h"

namespace ma {

template <class C>   // C: curve constants in the 5 x 51 field.c form (generated/curve_ED25519.h)
struct Ed26 {
    using F = Fe26;
    static constexpr uint32_t M26 = F::M26, M25 = F::M25;
    struct Ext { uint32_t X[10], Y[10], Z[10], T[10]; };

    // limbs below 2^31 -> tight (r[1] may keep a few units above 2^25); 32-bit registers only
    static MA_DEV void wc(uint32_t* f) {
        static_for<0, 9>([&](auto I) {
            constexpr int i = I;
            f[i + 1] += f[i] >> F::bits(i);
            f[i] &= (i & 1) ? M25 : M26;
        });
        const uint32_t c9 = f[9] >> 25;            // < 2^6
        f[9] &= M25;
        const uint32_t h0 = f[0] + 19u * c9;
        f[0] = h0 & M26;
        f[1] += h0 >> 26;
    }
    // 5 x 51-bit limbs (field.c form, limbs below 2^53: the contract of the curve layer) -> tight fe26
    static MA_DEV void from51(const spint* x, uint32_t* r) {
        static_for<0, 5>([&](auto K) {
            constexpr int k = K;
            r[2 * k] = (uint32_t)x[k] & M26;
            r[2 * k + 1] = (uint32_t)(x[k] >> 26);
        });
        wc(r);
    }
    static MA_DEV void d2(uint32_t* r) {            // 2d, folded at compile time from the curve constant
        spint d[5];
        static_for<0, 5>([&](auto I) { d[I] = C::b(I); });
        from51(d, r);
        F::add(r, r, r);
        wc(r);
    }

    // P = 2P, dbl-2008-hwcd with a = -1 (signs arranged so that no negation is needed):
    //   A = X^2, B = Y^2, C = 2Z^2, H = A + B, E = H - (X+Y)^2, G = A - B, F = C + G
    //   X3 = E F, Y3 = G H, Z3 = F G, T3 = E H.          Input X, Y, Z tight; T is not read.
    template <bool WANT_T>
    static MA_DEV void dbl(Ext& p) {
        uint32_t A[10], B[10], Cc[10], S[10], H[10], E[10], G[10], Ff[10], H19[10], F19[10];
        F::sqr(p.X, A);
        F::sqr(p.Y, B);
        F::sqr(p.Z, Cc);
        F::add(p.X, p.Y, S);        // 1.0
        F::sqr(S, S);
        F::add(A, B, H);            // 1.0
        F::sub(H, S, E);            // 2.0
        F::sub(A, B, G);            // 1.5
        F::add(Cc, Cc, Cc);         // 1.0
        F::add(Cc, G, Ff);          // 2.5
        wc(Ff);                     // tight
        F::pre19(Ff, F19);
        F::pre19(H, H19);
        F::mul(E, Ff, F19, p.X);    // 2.0 x 0.5
        F::mul(G, H, H19, p.Y);     // 1.5 x 1.0
        F::mul(G, Ff, F19, p.Z);    // 1.5 x 0.5
        if constexpr (WANT_T) F::mul(E, H, H19, p.T);
    }

    // the common tail of both additions: X3 = e f, Y3 = g h, Z3 = f g from a = (Y1-X1)(..), b = (Y1+X1)(..), c, d
    template <bool WANT_T = false>
    static MA_DEV void add_tail(const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d, Ext& p) {
        uint32_t e[10], f[10], g[10], h[10], g19[10], e19[10];
        F::sub(b, a, e);            // 1.5
        F::sub(d, c, f);            // 2.0
        F::add(d, c, g);            // 1.5
        F::add(b, a, h);            // 1.0
        F::pre19(g, g19);
        F::pre19(e, e19);
        F::mul(f, e, e19, p.X);     // 2.0 x 1.5
        F::mul(f, g, g19, p.Z);     // 2.0 x 1.5
        F::mul(h, g, g19, p.Y);     // 1.0 x 1.5
        if constexpr (WANT_T) F::mul(h, e, e19, p.T);   // 1.0 x 1.5
    }
    // P += Q, Q affine and cached as (y+x, y-x, 2dxy) (madd-2008-hwcd-3, a = -1, Z2 = 1); yp, ym tight, t2d <= 1.5.
    // Reads T of P; T of the sum is not produced (the next operation is a doubling, which does not read it).
    template <bool WANT_T = false>
    static MA_DEV void add_cached(Ext& p, const uint32_t* yp, const uint32_t* ym, const uint32_t* t2d) {
        uint32_t a[10], b[10], c[10], d[10];
        F::sub(p.Y, p.X, a);        // 1.5
        F::mul(a, ym, a);
        F::add(p.Y, p.X, b);        // 1.0
        F::mul(b, yp, b);
        F::mul(p.T, t2d, c);
        F::add(p.Z, p.Z, d);        // 1.0
        add_tail<WANT_T>(a, b, c, d, p);
    }
};

// (The window form of the single multiplication -- 3-bit signed windows, the table {1,2,3,4}P in registers / LDS, 255 doublings + 86
// mixed additions + two inversions per lane: rounds 2-4, 9.1e7/s, 170 spilled registers in its table builder -- was replaced in
// round 5 by the ladder form of csrc/ed26l.h: 1.13e8/s, no spills; same-box A/B in profiles/history/r05_lad_ab.log.)

// (The window form of the double multiplication -- 129 signed 2-bit windows, {P, 2P} in registers and {Q, 2Q} in LDS: rounds 2-4,
// 5.7e7/s, 178 spilled registers in its table builder -- was replaced in round 5 by the Straus form of csrc/ed26s.h: 7.9e7/s, no
// spills; same-box A/B in profiles/history/r05_mul2_ab.log.)

// Fused GENERATOR multiplication + affine export: the affine coordinates of e*G -- ecnXXXgen, ecnXXXmul, ecnXXXget, the
// opening of EdDSA key generation and signing (ed448.c:167-184 ED448_KEY_PAIR, 196-199 ED448_SIGN; curve.py builds the same
// layer for ED25519).  With the base point fixed there are no doublings: e' = e + sum_i 2^(W-1) 2^(W i), digit_i =
// window_i(e') - 2^(W-1), and e*G = sum_i digit_i * (2^(W i) G) with the NW x 2^(W-1) multiples precomputed in cached affine
// form (generated/comb_ED25519.h: W = 4, 65 windows x 8 entries, 62 400 bytes, one table for all lanes, read through
// wave-uniform addresses: TAB; with W = 5 the scan of 16 entries costs more than the 13 additions it saves).  Per window all its entries are read and selected by lane predication (a zero digit keeps the neutral element (1, 1, 0)),
// the sign swaps y+x / y-x and negates 2dxy, and one complete mixed addition (7M, T carried along) follows.
template <class C, class TAB, bool INIT = true>       // INIT = false: R += e*G (R holds a sum with its T coordinate)
MA_DEV void ed25519_mulgen_acc(const uint64_t* ew, typename Ed26<C>::Ext& R) {
    using E = Ed26<C>;
    using F = Fe26;
    constexpr int W = TAB::W, NW = TAB::NW, E2 = 1 << (W - 1);       // window width, windows, entries per window
    static_assert(W * NW >= 257 && W * NW <= 320, "e + bias must fit the windows and five words");
    uint64_t w[5];
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
        static_for<0, 5>([&](auto K) {
            constexpr int k = K;
            acc += (unsigned __int128)(k < 4 ? ew[k < 4 ? k : 0] : 0) + cw(k);
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
        static_for<0, 5>([&](auto K) {
            constexpr int k = K;
            w[k] >>= W;
            if constexpr (k < 4) w[k] |= w[k + 1] << (64 - W);
        });
        const bool neg = dgt < 0;
        const uint32_t m = (uint32_t)(neg ? -dgt : dgt);        // 0 .. 2^(W-1)
        uint32_t sel[3][10];
        static_for<0, 3>([&](auto CI) { static_for<0, 10>([&](auto K) { sel[CI][K] = (CI < 2 && K == 0) ? 1u : 0u; }); });
        // (selection as OR of masked entries: v_and_or_b32 reads the table entry straight from its scalar register, see wn26.h;
        // limb 0 of y+x and y-x starts at 1 only for a zero digit, so the OR never meets a set bit)
        static_for<0, 2>([&](auto CI) { sel[CI][0] = (m == 0) ? 1u : 0u; });
        static_for<0, E2>([&](auto MM) {
            constexpr int mm = MM;
            uint32_t mask = (m == (uint32_t)(mm + 1)) ? 0xffffffffu : 0u;
#if defined(__HIP_DEVICE_COMPILE__)
            asm("" : "+v"(mask));       // opaque: otherwise the compiler turns (entry & mask) back into a select with a move
#endif
            static_for<0, 3>([&](auto CI) {
                static_for<0, 10>([&](auto K) { sel[CI][K] |= (uint32_t)TAB::get(((i * E2 + mm) * 3 + CI) * 10 + K) & mask; });
            });
        });
        uint32_t yp[10], ym[10], nt[10];
        F::select(neg, sel[0], sel[1], yp);
        F::select(neg, sel[1], sel[0], ym);
        F::set(0, nt);
        F::sub(nt, sel[2], nt);                                 // 2p - t: 1.0 .. 1.5
        F::select(neg, sel[2], nt, sel[2]);
        E::template add_cached<true>(R, yp, ym, sel[2]);
    }
}
template <class C, class TAB>
MA_DEV void ed25519_mulgen_get_one(const uint64_t* ew, uint64_t* xw, uint64_t* yw) {
    using F = Fe26;
    typename Ed26<C>::Ext R;
    ed25519_mulgen_acc<C, TAB>(ew, R);
    uint32_t zi[10], ax[10], ay[10];
    F::invert(R.Z, zi);
    F::mul(R.X, zi, ax);
    F::mul(R.Y, zi, ay);
    F::to_words(ax, xw);
    F::to_words(ay, yw);
}
// rfc7748() on the BASE POINT u = 9 (public-key generation: every Diffie-Hellman exchange opens with it, rfc7748.c:297-333
// `rfc7748(alice, base, apk)`).  The generator of ED25519 is the image of (9, v) under the birational map u = (1 + y) / (1 - y),
// so [k](9) = (Z + Y) / (Z - Y) of k*G on the Edwards curve, with k*G from the fixed-base table (ed25519_mulgen_acc: no
// doublings, 65 mixed additions) instead of 255 ladder steps.  k is clamped as rfc7748.c:135-141 does; a clamped scalar is a
// multiple of 8 below 2^255 < 8q, so k*G is never the neutral element and Z - Y never 0.  G scalars per lane share one inversion.
template <class C, class TAB, int G, class LOAD>
MA_DEV void x25519_base_many(LOAD load, uint64_t (*ow)[4]) {
    using F = Fe26;
    uint32_t num[G][10], den[G][10], pre[G][10];
    typename Ed26<C>::Ext R;
#pragma unroll 1
    for (int g = 0; g < G; g++) {
        uint64_t kw[4];
        load(g, kw);
        kw[0] &= ~7ull;                                     // clamp (rfc7748.c:135-141)
        kw[3] = (kw[3] & 0x7fffffffffffffffull) | 0x4000000000000000ull;
        ed25519_mulgen_acc<C, TAB>(kw, R);
        static_for<0, G>([&](auto GI) {                     // (static register indices only)
            if (g == GI) { F::add(R.Z, R.Y, num[GI]); F::sub(R.Z, R.Y, den[GI]); }      // 1.0, 1.5
        });
    }
    F::copy(den[0], pre[0]);
    static_for<1, G>([&](auto GI) { F::mul(pre[GI - 1], den[GI], pre[GI]); });
    uint32_t inv[10], t[10], u[10];
    F::invert(pre[G - 1], inv);
    static_for<0, G>([&](auto GI) {
        constexpr int g = G - 1 - GI;
        if constexpr (g > 0) {
            F::mul(inv, pre[g - 1], t);
            F::mul(inv, den[g], inv);
        } else {
            F::copy(inv, t);
        }
        F::mul(num[g], t, u);
        F::to_words(u, ow[g]);
    });
}

}  // namespace ma
