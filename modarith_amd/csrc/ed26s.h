// modarith_amd/csrc/ed26s.h -- round 5: the fused ED25519 DOUBLE multiplication e*P + f*Q + affine export (ecnXXXmul2 followed by
// ecnXXXget, the verification pattern ed448.c:305 / nist256.c:251-254) as one Straus walk over signed 4-bit windows of both scalars.
//
// The reference's own mul2 (edwards.c:404-431, 486-510) is variable time -- a joint sparse form with data-dependent branches over
// public inputs (a signature, a public key) -- and only canonical affine bytes leave this kernel, so neither the constant-time
// table scan nor the reference's walk binds it.  Rounds 2-4 (ed26.h ed25519_mul2_get_dig) walked 129 signed 2-bit windows with the four
// entries {P, 2P, Q, 2Q} in registers / LDS: 258 doublings + 258 mixed additions + two inversions per lane = 3.6e5 multiply-adds,
// 178 registers spilled while the table was built.  Here:
//   * both scalars recoded into 65 signed 4-bit digits (e' = e + sum 8*16^i, digit = window - 8 in [-8, 7]): 256 doublings shared by both,
//     130 additions;
//   * the tables {0, 1, ..., 8}P and {0, ..., 8}Q as PROJECTIVE cached entries (Y+X, Y-X, 2dT, 2Z) -- no inversion in front of the walk
//     -- packed canonical, 128 bytes per entry = one cache line, in a per-wave slab of the workspace laid out lane-major;
//   * a lookup is ONE indexed load of the lane's own line (the digit is public): no scan, nothing of the table lives in registers;
//     the sign swaps Y+X / Y-X and negates 2dT by lane-predicated selects; control flow stays uniform;
//   * the affine export through the shared inversion of ed26l_k.h (one inversion per up to 32 records).
// 256 x (4S + 3M) + 65 M (T before an addition) + 65 x (8M + 7M) + 2 x 7 x 8M (tables) = 2.5e5 multiply-adds per pair.
#pragma once
#include "ed26.h"

namespace ma {

template <class C>
struct Ed26Straus {
    using F = Fe26;
    using E = Ed26<C>;
    using Ext = typename E::Ext;
    struct Cached { uint32_t yp[10], ym[10], t2d[10], z2[10]; };

    // p += q, q a projective cached point (add-2008-hwcd-3 with the second operand's sums, 2dT and 2Z precomputed; complete for a = -1)
    // want_t: a wave-uniform run-time flag (a loop counter, never data), so that ONE copy of the addition in the instruction stream
    // serves the sums that are added to again (T needed) and those that are doubled next (T not read)
    static MA_DEV void add_pc(Ext& p, const Cached& q, bool want_t) {
        uint32_t a[10], b[10], c[10], d[10], e[10], f[10], g[10], h[10], g19[10], e19[10];
        F::sub(p.Y, p.X, a);        // 1.5
        F::mul(a, q.ym, a);
        F::add(p.Y, p.X, b);        // 1.0
        F::mul(b, q.yp, b);
        F::mul(p.T, q.t2d, c);
        F::mul(p.Z, q.z2, d);
        F::sub(b, a, e);            // 1.5
        F::sub(d, c, f);            // 1.5
        F::add(d, c, g);            // 1.0
        F::add(b, a, h);            // 1.0
        F::pre19(g, g19);
        F::pre19(e, e19);
        F::mul(f, e, e19, p.X);
        F::mul(f, g, g19, p.Z);
        F::mul(h, g, g19, p.Y);
        if (want_t) F::mul(h, e, e19, p.T);
    }
    // the cached form of an extended point as four canonical packed elements (16 words)
    static MA_DEV void pack(const Ext& p, const uint32_t* dd, uint64_t (*w)[4]) {
        uint32_t s[10];
        F::add(p.Y, p.X, s);
        F::to_words(s, w[0]);
        F::sub(p.Y, p.X, s);
        F::to_words(s, w[1]);
        F::mul(p.T, dd, s);
        F::to_words(s, w[2]);
        F::add(p.Z, p.Z, s);
        F::to_words(s, w[3]);
    }
    // sign * entry: -(Y+X, Y-X, 2dT, 2Z) = (Y-X, Y+X, -2dT, 2Z)
    static MA_DEV void unpack(const uint64_t (*w)[4], bool neg, Cached& q) {
        uint64_t sp[4], sm[4];
        static_for<0, 4>([&](auto K) {
            const uint64_t a = w[0][K], b = w[1][K];
            sp[K] = neg ? b : a;
            sm[K] = neg ? a : b;
        });
        F::from_words(sp, q.yp);
        F::from_words(sm, q.ym);
        F::from_words(w[3], q.z2);
        uint32_t t[10], nt[10], zero[10];
        F::from_words(w[2], t);
        F::set(0, zero);
        F::sub(zero, t, nt);        // 2p - t: 1.0 .. 1.5
        F::select(neg, t, nt, q.t2d);
    }

    // entries 0 .. 8 of one table: k * (X : Y : Z), entry 0 the neutral element (1, 1, 0, 2).  One point is live at a time.
    template <class TAB>
    static MA_DEV void build(TAB& tab, int which, const spint* X, const spint* Y, const spint* Z) {
        uint32_t dd[10];
        E::d2(dd);
        Ext acc;
        Cached one;
        {
            uint32_t px[10], py[10], pz[10];
            E::from51(X, px);
            E::from51(Y, py);
            E::from51(Z, pz);
            F::mul(px, pz, acc.X);      // (XZ : YZ : Z^2 : XY)
            F::mul(py, pz, acc.Y);
            F::sqr(pz, acc.Z);
            F::mul(px, py, acc.T);
        }
        uint64_t w[4][4];
        static_for<0, 4>([&](auto CI) { static_for<0, 4>([&](auto K) { w[CI][K] = (K == 0) ? (CI < 2 ? 1u : (CI == 3 ? 2u : 0u)) : 0u; }); });
        tab.put(which, 0, w);
        pack(acc, dd, w);
        tab.put(which, 1, w);
        unpack(w, false, one);
#pragma unroll 1
        for (int k = 2; k <= 8; k++) {
            add_pc(acc, one, true);
            pack(acc, dd, w);
            tab.put(which, k, w);
        }
    }

    // R = e*P + f*Q.  de / df: window(i), i = 0 .. 64 from the top, of e' = e + sum 8*16^i (dig.window(i) - 8 = the signed digit).
    template <class DIG, class TAB>
    static MA_DEV void walk(DIG& de, DIG& df, TAB& tab, Ext& R) {
        F::set(0, R.X);
        F::set(1, R.Y);
        F::set(1, R.Z);
        F::set(0, R.T);
#pragma unroll 1
        for (int i = 0; i < 65; i++) {
            if (i != 0) {
                E::template dbl<false>(R);
                E::template dbl<false>(R);
                E::template dbl<false>(R);
                E::template dbl<true>(R);
            }
#pragma unroll 1
            for (int which = 0; which < 2; which++) {
                const int dgt = (int)(which ? df.window(i) : de.window(i)) - 8;       // [-8, 7]
                const bool neg = dgt < 0;
                const uint32_t m = (uint32_t)(neg ? -dgt : dgt);
                uint64_t w[4][4];
                tab.get(which, m, w);
                Cached q;
                unpack(w, neg, q);
                add_pc(R, q, which == 0);   // (T of the second sum is not read by the doublings)
            }
        }
    }
};

// e' = e + sum_{i<65} 8*16^i (260 bits), window 64 first
struct W25519_4Regs {
    uint64_t w[5];
    MA_DEV void init(const uint64_t* in) {
        unsigned __int128 acc = 0;
        uint64_t s[5];
        static_for<0, 5>([&](auto K) {
            constexpr int k = K;
            acc += (unsigned __int128)(k < 4 ? in[k < 4 ? k : 0] : 0) + (k < 4 ? 0x8888888888888888ull : 0x8ull);
            s[k] = (uint64_t)acc;
            acc >>= 64;
        });
        // left-align: bit 259 -> bit 63 of w[4]
        w[4] = (s[4] << 60) | (s[3] >> 4);
        w[3] = (s[3] << 60) | (s[2] >> 4);
        w[2] = (s[2] << 60) | (s[1] >> 4);
        w[1] = (s[1] << 60) | (s[0] >> 4);
        w[0] = s[0] << 60;
    }
    MA_DEV uint32_t window(int) {
        const uint32_t win = (uint32_t)(w[4] >> 60);
        w[4] = (w[4] << 4) | (w[3] >> 60);
        w[3] = (w[3] << 4) | (w[2] >> 60);
        w[2] = (w[2] << 4) | (w[1] >> 60);
        w[1] = (w[1] << 4) | (w[0] >> 60);
        w[0] <<= 4;
        return win;
    }
};
struct W25519_4Lds {                        // one byte per window in the lane's column of an LDS array
    const unsigned char* col;
    static MA_DEV void fill(const uint64_t* in, unsigned char* col) {
        W25519_4Regs r;
        r.init(in);
#pragma unroll 1
        for (int i = 0; i < 65; i++) col[(size_t)i * 64] = (unsigned char)r.window(i);
    }
    MA_DEV uint32_t window(int i) const { return col[(size_t)i * 64]; }
};

// the tables as a plain array (host check) ...
struct StrausTabArray {
    uint64_t t[2][9][16];
    MA_DEV void put(int which, int k, const uint64_t (*w)[4]) { static_for<0, 4>([&](auto CI) { static_for<0, 4>([&](auto K) { t[which][k][CI * 4 + K] = w[CI][K]; }); }); }
    MA_DEV void get(int which, uint32_t k, uint64_t (*w)[4]) const { static_for<0, 4>([&](auto CI) { static_for<0, 4>([&](auto K) { w[CI][K] = t[which][k][CI * 4 + K]; }); }); }
};
// ... and as the lane's 18 lines of a per-wave slab: entry (which, k) of lane l at slab + ((l * 18 + which * 9 + k) * 16 words
struct StrausTabSlab {
    uint64_t* lane;                         // slab + l * 18 * 16
    MA_DEV void put(int which, int k, const uint64_t (*w)[4]) const {
        uint64_t* p = lane + (size_t)(which * 9 + k) * 16;
        static_for<0, 4>([&](auto CI) { static_for<0, 4>([&](auto K) { p[CI * 4 + K] = w[CI][K]; }); });
    }
    MA_DEV void get(int which, uint32_t k, uint64_t (*w)[4]) const {
        const uint64_t* p = lane + (size_t)((uint32_t)which * 9u + k) * 16;
        static_for<0, 4>([&](auto CI) { static_for<0, 4>([&](auto K) { w[CI][K] = p[CI * 4 + K]; }); });
    }
};
constexpr size_t STRAUS_SLAB_BYTES_PER_WAVE = (size_t)64 * 18 * 16 * sizeof(uint64_t);      // 147 456

// one pair with its own inversion and plain-array tables: the per-lane reference of the kernel (tools/fe_host_check.hip)
template <class C>
MA_DEV void ed25519_mul2_get_straus_one(const uint64_t* ew, const spint* PX, const spint* PY, const spint* PZ,
                                        const uint64_t* fw, const spint* QX, const spint* QY, const spint* QZ, uint64_t* xw, uint64_t* yw) {
    using S = Ed26Straus<C>;
    using F = Fe26;
    StrausTabArray tab;
    S::build(tab, 0, PX, PY, PZ);
    S::build(tab, 1, QX, QY, QZ);
    W25519_4Regs de, df;
    de.init(ew);
    df.init(fw);
    typename S::Ext R;
    S::walk(de, df, tab, R);
    uint32_t zi[10], ax[10], ay[10];
    F::invert(R.Z, zi);
    F::mul(R.X, zi, ax);
    F::mul(R.Y, zi, ay);
    F::to_words(ax, xw);
    F::to_words(ay, yw);
}

}  // namespace ma
