// modarith_amd/csrc/glv26.h -- the secp256k1 endomorphism in the fused scalar multiplications (round 5).
//
// secp256k1 (y^2 = x^3 + 7 over a field with p = 1 mod 3; the curve curve.py:190-198 builds the Weierstrass layer for) carries
// phi(x, y) = (beta x, y), beta^3 = 1 in GF(p), and phi(P) = lambda P with lambda^3 = 1 mod the group order n (Gallant, Lambert,
// Vanstone, CRYPTO 2001).  A scalar splits as k = k1 + k2 lambda (mod n) with |k1|, |k2| < 2^128, so that
//     k P = k1 P + k2 phi(P)
// takes 128 doublings instead of 256 on ONE table {1..8}P (an entry of phi(P)'s table is the entry of P's with X multiplied by
// beta).  The reference has no such path (weierstrass.c:494-543 is a 256-bit fixed window); since ecnXXXmul is followed by
// ecnXXXget in the fused entry points, only the canonical affine bytes leave and any evaluation of k P gives the same ones --
// the same argument as for the window width and the limb form in wn26.h.  The group has prime order and cofactor 1: every point
// of the curve has order n (or is the point at infinity), so reducing the scalar mod n first changes nothing.
//
// The split is the lattice rounding of GLV with the short basis (a1, b1), (a2, b2) of {(x, y): x + y lambda = 0 mod n}:
//     c1 = round(k b2 / n), c2 = round(-k b1 / n), k1 = k - c1 a1 - c2 a2, k2 = -c1 b1 - c2 b2,
// the divisions by n replaced by multiplications with g1 = round(2^384 b2 / n), g2 = round(-2^384 b1 / n) and a shift by 384
// bits.  k1 + k2 lambda = k (mod n) holds by construction whatever c1, c2 are; the rounding only bounds the sizes: |k1|, |k2| <
// 2^128 (tests/test_host_arith.py checks identity and bound on 10^5 scalars against Python integers; the digit recoding below
// has room for 2^131).  Every step is word arithmetic on all lanes alike: masks, no branches on the scalar.
#pragma once
#include "wn26.h"

namespace ma {

struct GlvSecp256k1 {
    // constants as functions of a compile-time index (a constexpr array member would be a host variable to device code)
    static constexpr uint64_t n_(int i) { constexpr uint64_t v[4] = {0xBFD25E8CD0364141ull, 0xBAAEDCE6AF48A03Bull, 0xFFFFFFFFFFFFFFFEull, 0xFFFFFFFFFFFFFFFFull}; return v[i]; }
    static constexpr uint64_t g1_(int i) { constexpr uint64_t v[4] = {0xE893209A45DBB031ull, 0x3DAA8A1471E8CA7Full, 0xE86C90E49284EB15ull, 0x3086D221A7D46BCDull}; return v[i]; }
    static constexpr uint64_t g2_(int i) { constexpr uint64_t v[4] = {0x1571B4AE8AC47F71ull, 0x221208AC9DF506C6ull, 0x6F547FA90ABFE4C4ull, 0xE4437ED6010E8828ull}; return v[i]; }
    static constexpr uint64_t a1_(int i) { constexpr uint64_t v[3] = {0xE86C90E49284EB15ull, 0x3086D221A7D46BCDull, 0}; return v[i]; }          // a1 = b2
    static constexpr uint64_t a2_(int i) { constexpr uint64_t v[3] = {0x57C1108D9D44CFD8ull, 0x14CA50F7A8E2F3F6ull, 1}; return v[i]; }
    static constexpr uint64_t mb1_(int i) { constexpr uint64_t v[3] = {0x6F547FA90ABFE4C3ull, 0xE4437ED6010E8828ull, 0}; return v[i]; }         // -b1
    static constexpr uint64_t beta_(int i) { constexpr uint64_t v[4] = {0xC1396C28719501EEull, 0x9CF0497512F58995ull, 0x6E64479EAC3434E9ull, 0x7AE96A2B657C0710ull}; return v[i]; }

    // limb i of beta in the fk26 form (26 bits each, canonical)
    static constexpr int32_t beta26(int i) {
        const int o = 26 * i, wi = o / 64, sh = o % 64;
        uint64_t v = beta_(wi) >> sh;
        if (sh + 26 > 64 && wi + 1 < 4) v |= beta_(wi + 1) << (64 - sh);
        return (int32_t)(v & ((1u << 26) - 1));
    }

    // words 6, 7 of k g + 2^383: round(k g / 2^384)
    static MA_DEV void round_mul(const uint64_t* k, const uint64_t* g, uint64_t* c) {
        uint64_t p[8];
        static_for<0, 8>([&](auto I) { p[I] = 0; });
        static_for<0, 4>([&](auto II) {
            constexpr int i = II;
            uint64_t carry = 0;
            static_for<0, 4>([&](auto JJ) {
                constexpr int j = JJ;
                const unsigned __int128 t = (unsigned __int128)k[i] * g[j] + p[i + j] + carry;
                p[i + j] = (uint64_t)t;
                carry = (uint64_t)(t >> 64);
            });
            p[i + 4] = carry;
        });
        const unsigned __int128 t5 = (unsigned __int128)p[5] + ((uint64_t)1 << 63);
        const unsigned __int128 t6 = (unsigned __int128)p[6] + (uint64_t)(t5 >> 64);
        c[0] = (uint64_t)t6;
        c[1] = p[7] + (uint64_t)(t6 >> 64);
    }
    // the low 192 bits of a (two words) times b (three words)
    static MA_DEV void mul_lo3(const uint64_t* a, const uint64_t* b, uint64_t* r) {
        const unsigned __int128 t0 = (unsigned __int128)a[0] * b[0];
        const unsigned __int128 t1 = (unsigned __int128)a[0] * b[1] + (uint64_t)(t0 >> 64);
        const unsigned __int128 t2 = (unsigned __int128)a[1] * b[0] + (uint64_t)t1;
        r[0] = (uint64_t)t0;
        r[1] = (uint64_t)t2;
        r[2] = a[0] * b[2] + a[1] * b[1] + (uint64_t)(t1 >> 64) + (uint64_t)(t2 >> 64);
    }
    static MA_DEV void sub3(const uint64_t* a, const uint64_t* b, uint64_t* r) {
        const uint64_t d0 = a[0] - b[0], bw0 = a[0] < b[0];
        const uint64_t d1 = a[1] - b[1], bw1 = (a[1] < b[1]) | ((d1 < bw0) ? 1u : 0u);
        r[0] = d0;
        r[1] = d1 - bw0;
        r[2] = a[2] - b[2] - bw1;
    }
    // r = |v| for a 192-bit two's-complement v; returns v < 0
    static MA_DEV bool abs3(uint64_t* v) {
        const uint64_t m = (uint64_t)0 - (v[2] >> 63);
        uint64_t x[3] = {v[0] ^ m, v[1] ^ m, v[2] ^ m}, mm[3] = {m, m, m};
        sub3(x, mm, v);
        return m != 0;
    }

    // e (four little-endian words, any 256-bit value) -> |k1|, |k2| (three words each, below 2^129) and their signs
    static MA_DEV void split(const uint64_t* ew, uint64_t* k1, bool& n1, uint64_t* k2, bool& n2) {
        uint64_t k[4];
        {   // k = e mod n: e < 2^256 < 2n, one subtraction
            uint64_t d[4], bw = 0;
            static_for<0, 4>([&](auto I) {
                constexpr uint64_t nI = n_(I);
                const uint64_t x = ew[I] - nI, b1 = ew[I] < nI;
                d[I] = x - bw;
                bw = b1 | ((x < bw) ? 1u : 0u);
            });
            const uint64_t keep = (uint64_t)0 - bw;             // borrow: e < n, keep e
            static_for<0, 4>([&](auto I) { k[I] = (ew[I] & keep) | (d[I] & ~keep); });
        }
        uint64_t c1[2], c2[2], t[3], u[3], G1[4], G2[4], A1[3], A2[3], MB1[3];
        static_for<0, 4>([&](auto I) { G1[I] = g1_(I); G2[I] = g2_(I); });
        static_for<0, 3>([&](auto I) { A1[I] = a1_(I); A2[I] = a2_(I); MB1[I] = mb1_(I); });
        round_mul(k, G1, c1);
        round_mul(k, G2, c2);
        mul_lo3(c1, A1, t);
        sub3(k, t, k1);
        mul_lo3(c2, A2, t);
        sub3(k1, t, k1);                                        // k - c1 a1 - c2 a2  (mod 2^192; the value is below 2^128 in size)
        mul_lo3(c1, MB1, t);
        mul_lo3(c2, A1, u);
        sub3(t, u, k2);                                         // -c1 b1 - c2 b2
        n1 = abs3(k1);
        n2 = abs3(k2);
    }
};

// signed 4-bit digits of a magnitude below 2^131 (three words): s = k + sum_{i<33} 8 * 16^i, digit_i = window_i(s) - 8, windows
// taken from the top (wn26_recode for 132 bits in three words)
struct Glv4Regs {
    uint64_t w[3];
    MA_DEV void init(const uint64_t* k) {
        constexpr uint64_t B = 0x8888888888888888ull;
        const unsigned __int128 a0 = (unsigned __int128)k[0] + B;
        const unsigned __int128 a1 = (unsigned __int128)k[1] + B + (uint64_t)(a0 >> 64);
        const uint64_t s2 = k[2] + 0x8u + (uint64_t)(a1 >> 64);
        const uint64_t s0 = (uint64_t)a0, s1 = (uint64_t)a1;
        w[2] = (s2 << 60) | (s1 >> 4);
        w[1] = (s1 << 60) | (s0 >> 4);
        w[0] = s0 << 60;
    }
    MA_DEV uint32_t take() {
        const uint32_t win = (uint32_t)(w[2] >> 60);
        w[2] = (w[2] << 4) | (w[1] >> 60);
        w[1] = (w[1] << 4) | (w[0] >> 60);
        w[0] <<= 4;
        return win;
    }
};
constexpr int GLV_WINDOWS = 33;

// digit sources for the two half scalars: window(which, i) for i = 0, 1, ... in order, neg(which) = the half scalar's sign
struct GlvRegs {
    Glv4Regs r[2];
    bool n[2];
    MA_DEV void init(const uint64_t* ew) {
        uint64_t k1[3], k2[3];
        GlvSecp256k1::split(ew, k1, n[0], k2, n[1]);
        r[0].init(k1);
        r[1].init(k2);
    }
    MA_DEV uint32_t window(int which, int) { return which ? r[1].take() : r[0].take(); }
    MA_DEV bool neg(int which) const { return which ? n[1] : n[0]; }
};
struct GlvLds {
    static constexpr int COUNT = 2 * GLV_WINDOWS;               // bytes per lane column
    const unsigned char* col;
    bool n[2];
    // one byte per window in the lane's column of an LDS array (as wn26.h WnLds), written before the point is loaded
    MA_DEV void fill(const uint64_t* ew, unsigned char* c) {
        uint64_t k1[3], k2[3];
        GlvSecp256k1::split(ew, k1, n[0], k2, n[1]);
        Glv4Regs r;
        r.init(k1);
#pragma unroll 1
        for (int i = 0; i < GLV_WINDOWS; i++) c[(size_t)i * 64] = (unsigned char)r.take();
        r.init(k2);
#pragma unroll 1
        for (int i = 0; i < GLV_WINDOWS; i++) c[(size_t)(GLV_WINDOWS + i) * 64] = (unsigned char)r.take();
        col = c;
    }
    MA_DEV uint32_t window(int which, int i) const { return col[(size_t)(which * GLV_WINDOWS + i) * 64]; }
    MA_DEV bool neg(int which) const { return which ? n[1] : n[0]; }
};

// R = k P from the digits of k's two halves: 33 windows of (four doublings, +- table[|d1|], +- phi(table[|d2|])) on the one table
// {1..8}P of wn26.h (per-lane slot, every lookup reads all eight entries).  128 doublings + 66 additions + 33 multiplications by
// beta + the table (4 + 3) against 256 + 65 + table of wn26_mul_acc.
template <class TAB, class DIG>
MA_DEV void secp256k1_glv_mul_acc(DIG& dig, const spint* X, const spint* Y, const spint* Z, const TAB& T, Wn26<CvSecp256k1>::Pt& R) {
    using E = Wn26<CvSecp256k1>;
    using F = Fk26;
    E::Pt Q;
    E::load_point(X, Y, Z, Q);
    E::template build_table<8>(Q, T.origin(), T.stride(), 0);
    E::inf(R);
#pragma unroll 1
    for (int i = 0; i < GLV_WINDOWS; i++) {
        if (i != 0) {
#pragma unroll 1
            for (int j = 0; j < 4; j++) E::dbl(R);
        }
#pragma unroll 1
        for (int which = 0; which < 2; which++) {
            const int dgt = (int)dig.window(which, i) - 8;                  // [-8, 7]
            const bool dn = dgt < 0;
            const uint32_t m = (uint32_t)(dn ? -dgt : dgt);                 // 0..8
            E::template lookup<8>(T.origin(), T.stride(), 0, m, dn != dig.neg(which), Q);
            if (which) {                                                    // phi: X *= beta (uniform over the wave)
                int32_t b[10];
                static_for<0, 10>([&](auto I) { b[I] = GlvSecp256k1::beta26(I); });
                F::mul(Q.X, b, Q.X);
            }
            E::add(Q, R);
        }
    }
}
template <class TAB, class DIG>
MA_DEV void secp256k1_glv_mul_get_dig(DIG& dig, const spint* X, const spint* Y, const spint* Z, const TAB& T, uint64_t* xw, uint64_t* yw) {
    Wn26<CvSecp256k1>::Pt R;
    secp256k1_glv_mul_acc(dig, X, Y, Z, T, R);
    Wn26<CvSecp256k1>::affine_words(R, xw, yw);
}
// e G + f Q: f Q as above, e G through the fixed-base table (wn26_mulgen_acc, no doublings of its own)
template <class COMB, class TAB, class DIG>
MA_DEV void secp256k1_glv_mulgen2_get_dig(const uint64_t* ew, DIG& digf, const spint* QX, const spint* QY, const spint* QZ, const TAB& T,
                                          uint64_t* xw, uint64_t* yw) {
    Wn26<CvSecp256k1>::Pt R;
    secp256k1_glv_mul_acc(digf, QX, QY, QZ, T, R);
    wn26_mulgen_acc<CvSecp256k1, COMB, false>(ew, R);
    Wn26<CvSecp256k1>::affine_words(R, xw, yw);
}
template <class COMB, class TAB, class DIG>
MA_DEV void secp256k1_glv_mulgen2_acc(const uint64_t* ew, DIG& digf, const spint* QX, const spint* QY, const spint* QZ, const TAB& T,
                                      Wn26<CvSecp256k1>::Pt& R) {
    secp256k1_glv_mul_acc(digf, QX, QY, QZ, T, R);
    wn26_mulgen_acc<CvSecp256k1, COMB, false>(ew, R);
}

// e P + f Q: both scalars split, the tables {1..8}P and {1..8}Q in a double slot (entries 0..7 and 8..15), per window four doublings
// and four additions: 128 doublings + 132 additions against the 255 + 172 of wn26_mul2_get_dig (three-bit windows on two tables of
// four).  Every lookup scans its eight entries.  An infinite result leaves as (0, 1).
constexpr int GLV2_TABLE_WORDS = 2 * WN26_TABLE_WORDS;
// loadP(X, Y, Z) / loadQ fetch the 3 x 5 limbs of a point when its table is about to be built (the kernel: from the caller's arrays -- Q's
// thirty registers are then not live while P's table is built)
template <class TAB, class DIG, class LP, class LQ>
MA_DEV void secp256k1_glv_mul2_acc_ld(DIG& dige, LP loadP, DIG& digf, LQ loadQ, const TAB& T, Wn26<CvSecp256k1>::Pt& R) {
    using E = Wn26<CvSecp256k1>;
    using F = Fk26;
    E::Pt Q;
    {
        spint X[5], Y[5], Z[5];
        loadP(X, Y, Z);
        E::load_point(X, Y, Z, Q);
        E::template build_table<8>(Q, T.origin(), T.stride(), 0);
        loadQ(X, Y, Z);
        E::load_point(X, Y, Z, Q);
        E::template build_table<8>(Q, T.origin(), T.stride(), 8);
    }
    E::inf(R);
#pragma unroll 1
    for (int i = 0; i < GLV_WINDOWS; i++) {
        if (i != 0) {
#pragma unroll 1
            for (int j = 0; j < 4; j++) E::dbl(R);
        }
#pragma unroll 1
        for (int which = 0; which < 4; which++) {
            const int pt = which >> 1, half = which & 1;
            const int dgt = (int)(pt ? digf.window(half, i) : dige.window(half, i)) - 8;
            const bool dn = dgt < 0, sn = pt ? digf.neg(half) : dige.neg(half);
            const uint32_t m = (uint32_t)(dn ? -dgt : dgt);
            E::template lookup<8>(T.origin(), T.stride(), 8 * pt, m, dn != sn, Q);
            if (half) {
                int32_t b[10];
                static_for<0, 10>([&](auto I) { b[I] = GlvSecp256k1::beta26(I); });
                F::mul(Q.X, b, Q.X);
            }
            E::add(Q, R);
        }
    }
}
template <class TAB, class DIG>
MA_DEV void secp256k1_glv_mul2_acc(DIG& dige, const spint* PX, const spint* PY, const spint* PZ,
                                   DIG& digf, const spint* QX, const spint* QY, const spint* QZ, const TAB& T, Wn26<CvSecp256k1>::Pt& R) {
    auto cp = [](const spint* a, const spint* b, const spint* c) {
        return [=](spint* x, spint* y, spint* z) { static_for<0, 5>([&](auto I) { x[I] = a[I]; y[I] = b[I]; z[I] = c[I]; }); };
    };
    secp256k1_glv_mul2_acc_ld(dige, cp(PX, PY, PZ), digf, cp(QX, QY, QZ), T, R);
}
template <class TAB, class DIG>
MA_DEV void secp256k1_glv_mul2_get_dig(DIG& dige, const spint* PX, const spint* PY, const spint* PZ,
                                       DIG& digf, const spint* QX, const spint* QY, const spint* QZ, const TAB& T, uint64_t* xw, uint64_t* yw) {
    Wn26<CvSecp256k1>::Pt R;
    secp256k1_glv_mul2_acc(dige, PX, PY, PZ, digf, QX, QY, QZ, T, R);
    Wn26<CvSecp256k1>::affine_words(R, xw, yw);
}

}  // namespace ma
