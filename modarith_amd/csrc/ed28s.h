// modarith_amd/csrc/ed28s.h -- round 5: the fused ED448 DOUBLE multiplication e*P + f*Q + affine export (ecnXXXmul2 followed by ecnXXXget,
// ED448_VERIFY's pattern ed448.c:290-310) as one Straus walk over signed 4-bit windows of both scalars: the construction of
// csrc/ed26s.h (which see) on the fe28 representation.
//
// Rounds 2-4 (ed28.h ed448_mul2_get_one) walked 225 signed 2-bit windows with the affine tables {P, 2P}, {Q, 2Q} scanned in a workspace
// slab: 450 doublings + 450 mixed additions + two inversions per pair.  Here both scalars are recoded into 113 signed 4-bit digits
// (e' = e + sum 8*16^i < 2^452): 448 doublings shared by both, 226 additions of PROJECTIVE table entries (X, Y, 39081 T, Z) -- no
// inversion in front of the walk --, the tables {0, ..., 8}P and {0, ..., 8}Q packed canonical, 4 x 56 bytes per entry in two cache lines
// of a lane-major per-wave slab, read BY INDEX (the inputs of a verification are public, the reference's own mul2 is variable time:
// edwards.c:404-431).  An entry's coordinates are fetched one at a time where the addition uses them, so the point being added never
// sits in registers next to the accumulator; the sign negates X and 39081 T by selects; control flow stays uniform.  The affine export
// goes through the shared inversion of csrc/edlad_k.h.
#pragma once
#include "ed28.h"

namespace ma {

struct Ed28Straus {
    using F = Fe28;
    using E = Ed28;
    using Ext = E::Ext;

    // p += sign * q, q = (X, Y, 39081 T, Z) fetched coordinate by coordinate: fetch(c, out) -> tight limbs of coordinate c.
    // (add-2008-hwcd for a = 1, d = -39081: ed28.h add_tail.)  want_t: a wave-uniform flag, one copy of the addition in the instruction stream.
    template <class FETCH>
    static MA_DEV void add_pc(Ext& p, FETCH fetch, bool neg, bool want_t) {
        uint32_t A[16], B[16], Cc[16], D[16], M[16];
        {
            uint32_t q[16], nq[16];
            fetch(2, q);                        // 39081 T
            E::neg2p(q, nq);                    // below 2^29
            F::select(neg, q, nq, q);
            F::mul_k(p.T, q, Cc);
            fetch(3, q);
            F::mul_k(p.Z, q, D);
        }
        {
            uint32_t qx[16], qy[16], nx[16], s1[16], s2[16];
            fetch(0, qx);
            E::neg2p(qx, nx);
            F::select(neg, qx, nx, qx);         // below 2^29
            fetch(1, qy);
            F::mul_k(p.X, qx, A);
            F::mul_k(p.Y, qy, B);
            F::add(p.X, p.Y, s1);               // below 2^29
            F::add(qx, qy, s2);                 // below 1.5 * 2^29
            F::mul(s1, s2, M);
        }
        E::add_tail(A, B, Cc, D, M, p, want_t);
    }

    // entries 0 .. 8 of one table: k * (X : Y : Z); entry 0 the neutral element (0, 1, 0, 1).  put(which, k, c, words): coordinate c
    // of entry k as seven canonical words.  One point is live at a time; P itself is re-read from entry 1.
    template <class TAB>
    static MA_DEV void build(TAB& tab, int which, const spint* X, const spint* Y, const spint* Z) {
        Ext acc;
        {
            uint32_t px[16], py[16], pz[16];
            E::from56(X, px);
            E::from56(Y, py);
            E::from56(Z, pz);
            F::mul_k(px, pz, acc.X);            // (XZ : YZ : Z^2 : XY)
            F::mul_k(py, pz, acc.Y);
            F::sqr_k(pz, acc.Z);
            F::mul_k(px, py, acc.T);
        }
        uint64_t w[7];
        static_for<0, 4>([&](auto CI) {
            static_for<0, 7>([&](auto K) { w[K] = (K == 0 && (CI == 1 || CI == 3)) ? 1u : 0u; });
            tab.put(which, 0, CI, w);
        });
        auto store = [&](int k) {
            uint32_t td[16];
            F::to_words(acc.X, w);
            tab.put(which, k, 0, w);
            F::to_words(acc.Y, w);
            tab.put(which, k, 1, w);
            F::template mul_small<E::D_ABS>(acc.T, td);
            F::to_words(td, w);
            tab.put(which, k, 2, w);
            F::to_words(acc.Z, w);
            tab.put(which, k, 3, w);
        };
        store(1);
#pragma unroll 1
        for (int k = 2; k <= 8; k++) {
            add_pc(acc, [&](int c, uint32_t* o) { uint64_t v[7]; tab.get(which, 1, c, v); F::from_words(v, o); }, false, true);
            store(k);
        }
    }

    // R = e*P + f*Q.  de / df: window(i), i = 0 .. 112 from the top, of e' = e + sum 8*16^i (window - 8 = the signed digit)
    template <class DIG, class TAB>
    static MA_DEV void walk(DIG& de, DIG& df, TAB& tab, Ext& R) {
        F::set(0, R.X);
        F::set(1, R.Y);
        F::set(1, R.Z);
        F::set(0, R.T);
#pragma unroll 1
        for (int i = 0; i < 113; i++) {
            if (i != 0) {
#pragma unroll 1
                for (int k = 0; k < 4; k++) E::dbl(R, k == 3);
            }
#pragma unroll 1
            for (int which = 0; which < 2; which++) {
                const int dgt = (int)(which ? df.window(i) : de.window(i)) - 8;       // [-8, 7]
                const bool neg = dgt < 0;
                const uint32_t m = (uint32_t)(neg ? -dgt : dgt);
                add_pc(R, [&](int c, uint32_t* o) { uint64_t v[7]; tab.get(which, m, c, v); F::from_words(v, o); }, neg, which == 0);
            }
        }
    }
};

// e' = e + sum_{i<113} 8*16^i (452 bits), window 112 first
struct W448_4Regs {
    uint64_t w[8];
    MA_DEV void init(const uint64_t* in) {
        unsigned __int128 acc = 0;
        uint64_t s[8];
        static_for<0, 8>([&](auto K) {
            constexpr int k = K;
            acc += (unsigned __int128)(k < 7 ? in[k < 7 ? k : 0] : 0) + (k < 7 ? 0x8888888888888888ull : 0x8ull);
            s[k] = (uint64_t)acc;
            acc >>= 64;
        });
        // left-align: bit 451 -> bit 63 of w[7]
        static_for<0, 8>([&](auto KK) {
            constexpr int k = 7 - KK;
            w[k] = (s[k] << 60) | (k > 0 ? s[k > 0 ? k - 1 : 0] >> 4 : 0);
        });
    }
    MA_DEV uint32_t window(int) {
        const uint32_t win = (uint32_t)(w[7] >> 60);
        static_for<0, 8>([&](auto KK) {
            constexpr int k = 7 - KK;
            w[k] = (w[k] << 4) | (k > 0 ? w[k > 0 ? k - 1 : 0] >> 60 : 0);
        });
        return win;
    }
};
struct W448_4Lds {                          // one byte per window in the lane's column of an LDS array
    const unsigned char* col;
    static MA_DEV void fill(const uint64_t* in, unsigned char* col) {
        W448_4Regs r;
        r.init(in);
#pragma unroll 1
        for (int i = 0; i < 113; i++) col[(size_t)i * 64] = (unsigned char)r.window(i);
    }
    MA_DEV uint32_t window(int i) const { return col[(size_t)i * 64]; }
};

// the tables as a plain array (host check) ...
struct Straus448TabArray {
    uint64_t t[2][9][4][7];
    MA_DEV void put(int which, int k, int c, const uint64_t* w) { static_for<0, 7>([&](auto K) { t[which][k][c][K] = w[K]; }); }
    MA_DEV void get(int which, uint32_t k, int c, uint64_t* w) const { static_for<0, 7>([&](auto K) { w[K] = t[which][k][c][K]; }); }
};
// ... and as the lane's 18 entries of a per-wave slab, 32 words (two cache lines) each: coordinate c of entry (which, k) of lane l at
// slab + ((l * 18 + which * 9 + k) * 32 + c * 8 words (seven used)
struct Straus448TabSlab {
    uint64_t* lane;
    MA_DEV void put(int which, int k, int c, const uint64_t* w) const {
        uint64_t* p = lane + (size_t)(which * 9 + k) * 32 + c * 8;
        static_for<0, 7>([&](auto K) { p[K] = w[K]; });
    }
    MA_DEV void get(int which, uint32_t k, int c, uint64_t* w) const {
        const uint64_t* p = lane + (size_t)((uint32_t)which * 9u + k) * 32 + c * 8;
        static_for<0, 7>([&](auto K) { w[K] = p[K]; });
    }
};
constexpr size_t STRAUS448_SLAB_BYTES_PER_WAVE = (size_t)64 * 18 * 32 * sizeof(uint64_t);       // 294 912

// one pair with its own inversion and plain-array tables: the per-lane reference of the kernel (tools/fe_host_check.hip)
MA_DEV void ed448_mul2_get_straus_one(const uint64_t* ew, const spint* PX, const spint* PY, const spint* PZ,
                                      const uint64_t* fw, const spint* QX, const spint* QY, const spint* QZ, uint64_t* xw, uint64_t* yw) {
    using S = Ed28Straus;
    using F = Fe28;
    Straus448TabArray tab;
    S::build(tab, 0, PX, PY, PZ);
    S::build(tab, 1, QX, QY, QZ);
    W448_4Regs de, df;
    de.init(ew);
    df.init(fw);
    S::Ext R;
    S::walk(de, df, tab, R);
    uint32_t zi[16], ax[16], ay[16];
    F::invert(R.Z, zi);
    F::mul_k(R.X, zi, ax);
    F::mul_k(R.Y, zi, ay);
    F::to_words(ax, xw);
    F::to_words(ay, yw);
}

}  // namespace ma
