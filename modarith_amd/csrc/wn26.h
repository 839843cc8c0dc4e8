// modarith_amd/csrc/wn26.h -- fused ecnXXXmul + ecnXXXget (and ecnXXXmul2 + ecnXXXget) for NIST P-256 on the fm26
// representation (gfx950).
//
// The reference's ECDSA code ends every scalar multiplication in ecnXXXget: key generation and signing
// (nist256.c:155-161, 219-222: ecnXXXmul then ecnXXXget), verification (nist256.c:251-256: ecnXXXmul2, ecnXXXisinf,
// ecnXXXget).  Only canonical big-endian coordinate bytes leave, so -- exactly as for ED25519 in ed26.h -- the limb form
// and the window width are free; the group law is kept: the COMPLETE projective formulas of Renes-Costello-Batina for
// a = -3 (eprint 2015/1060 algorithms 4 and 6, the ones weierstrass.c:68-281 implements), which have no exceptional
// cases on a curve of prime order, so the affine point reached is the reference's for every input point ON the curve,
// the point at infinity included (it leaves as x = 0, y = 1, the bytes ecnXXXget gives after weierstrass.c:299-310).
// (For off-curve input neither side means anything; they may differ.)
//
// Scalar multiplication is fixed-window as weierstrass.c:494-543 -- the same instruction and address sequence for every
// scalar -- with signed 4-bit digits from a carry-free recoding: e' = e + sum_{i<65} 8*16^i < 16^65, digit_i =
// window_i(e') - 8 in [-8, 7].  The table {1..8}P (projective, fm26 limbs as computed, 15 words per entry) lives in a
// per-lane slot of a global workspace [entry][word][lane] and every lookup reads all eight entries with lane-predicated
// selects; the digit's sign negates Y.  Work per scalar: 256 doublings + 65 additions + table (4 + 3) + one inversion.
//
// K (fm26.h) is given in the comments: |limb| <= K 2^26.  Point coordinates entering add / dbl have K <= 4.
//
// secp256k1 (a = 0, b = 7; the second Weierstrass curve of curve.py:190-198 with a 256-bit pseudo-Mersenne field) runs the
// same scalar-multiplication code on fk26.h with the a = 0 formulas (RCB algorithms 7 and 9, weierstrass.c:120-157,
// 189-226): 3b = 21 is a small constant, the doubling costs 6M + 2S instead of 8M + 3S + 2m_b.
//
// Round 5: the P-256 KERNELS run csrc/wj26.h (Jacobian coordinates on affine window tables) on top of this file's point type, complete
// additions (last window, generator part), lookups and export; the secp256k1 kernels run this file's complete formulas under the
// scalar split of csrc/glv26.h.  wn26_mul_get_one / wn26_mul2_get_one below remain the complete-formula statement of both curves that
// the host checks (tools/fe_host_check.hip, tools/wn26_host.hip) hold against the oracle.
#pragma once
#include "fm26.h"
#include "fk26.h"

namespace ma {

constexpr int WN26_TABLE_WORDS = 8 * 15;        // 64-bit words per lane slot: eight entries of (X, Y, Z), two limbs per word
constexpr int NIST256_TABLE_WORDS = WN26_TABLE_WORDS;

// curve policies: the field representation, a, and the multiplication by the curve constant
struct CvNist256 {
    using F = Fm26;
    static constexpr int A = -3;
    // b 2^286 mod p
    static constexpr int32_t bhat(int i) {
        constexpr int32_t v[10] = {0x30c0187, 0x4bddfd, 0x37d88a7, 0x274d89c, 0x327150a, 0x2cf005c, 0x84bb5a, 0x21a8ff7, 0x394025c, 0x1e0b74};
        return v[i];
    }
    static MA_DEV void mulb(const int32_t* f, int32_t* r) {
        int32_t b[10];
        static_for<0, 10>([&](auto I) { b[I] = bhat(I); });
        F::mul(f, b, r);
    }
};
struct CvSecp256k1 {
    using F = Fk26;
    static constexpr int A = 0;
    static MA_DEV void mul3b(const int32_t* f, int32_t* r) { F::template mul_small<21>(f, r); }   // 3b = 21
};

template <class CV>
struct Wn26 {
    using F = typename CV::F;
    struct Pt { int32_t X[10], Y[10], Z[10]; };

    static MA_DEV void mulb(const int32_t* f, int32_t* r) { CV::mulb(f, r); }
    static MA_DEV void inf(Pt& p) { F::zero(p.X); F::set_one(p.Y); F::zero(p.Z); }

    // a = 0: P += Q (RCB algorithm 7; weierstrass.c:120-157).  Inputs K <= 2; X3, Y3, Z3 each one fold of two products.
    static MA_DEV void add0(const Pt& q, Pt& p) {
        int32_t T0[10], T1[10], T2[10], T3[10], T4[10], U[10], V[10];
        F::mul(p.X, q.X, T0);
        F::mul(p.Y, q.Y, T1);
        F::mul(p.Z, q.Z, T2);
        F::add(p.X, p.Y, T3);       // 4
        F::add(q.X, q.Y, T4);       // 4
        F::mul(T3, T4, T3);
        F::add(T0, T1, T4);         // 2
        F::sub(T3, T4, T3);         // 3   X1Y2 + X2Y1
        F::add(p.Y, p.Z, T4);       // 4
        F::add(q.Y, q.Z, U);        // 4
        F::mul(T4, U, T4);
        F::add(T1, T2, U);          // 2
        F::sub(T4, U, T4);          // 3   Y1Z2 + Y2Z1
        F::add(p.X, p.Z, U);        // 4
        F::add(q.Z, q.X, V);        // 4
        F::mul(U, V, U);
        F::add(T0, T2, V);          // 2
        F::sub(U, V, U);            // 3   X1Z2 + X2Z1
        F::add(T0, T0, V);          // 2
        F::add(T0, V, T0);          // 3   3 X1X2
        CV::mul3b(T2, T2);          // 1   3b Z1Z2
        CV::mul3b(U, U);            // 1   3b (X1Z2 + X2Z1)
        F::add(T1, T2, V);          // 2   Y1Y2 + 3b Z1Z2
        F::sub(T1, T2, T1);         // 2   Y1Y2 - 3b Z1Z2
        F::neg(U, T2);              // 1
        F::mul2(T3, T1, T2, T4, p.X);    // X3 = T3 T1 - U T4:   3 x 2 + 1 x 3
        F::mul2(U, T0, T1, V, p.Y);      // Y3 = U T0 + T1 V:    1 x 3 + 2 x 2
        F::mul2(V, T4, T0, T3, p.Z);     // Z3 = V T4 + T0 T3:   2 x 3 + 3 x 3
    }
    // a = 0: P = 2P (RCB algorithm 9; weierstrass.c:189-226).  Outputs: X K = 2, Y, Z K = 1.
    static MA_DEV void dbl0(Pt& p) {
        int32_t T0[10], T1[10], T2[10], T3[10], T4[10], U[10];
        F::sqr(p.Y, T0);
        F::add(T0, T0, T3);         // 2
        F::add(T3, T3, T3);         // 4
        F::add(T3, T3, T3);         // 8   8 Y^2
        F::mul(p.X, p.Y, T4);
        F::mul(p.Y, p.Z, T1);
        F::sqr(p.Z, T2);
        CV::mul3b(T2, T2);          // 1   3b Z^2
        F::add(T0, T2, U);          // 2   Y^2 + 3b Z^2
        F::mul(T3, T1, p.Z);        // Z3 = 8 Y^3 Z
        F::add(T2, T2, T1);         // 2
        F::add(T2, T1, T1);         // 3   9b Z^2
        F::sub(T0, T1, T0);         // 4   Y^2 - 9b Z^2
        F::mul2(U, T0, T2, T3, p.Y);     // Y3 = (Y^2 + 3b Z^2)(Y^2 - 9b Z^2) + 3b Z^2 8 Y^2:  2 x 4 + 1 x 8
        F::mul(T0, T4, p.X);
        F::add(p.X, p.X, p.X);      // 2   X3 = 2 X Y (Y^2 - 9b Z^2)
    }

    // P += Q (RCB algorithm 4, a = -3; operation order of weierstrass.c:68-175)
    static MA_DEV void add(const Pt& q, Pt& p) {
        if constexpr (CV::A == 0) { add0(q, p); return; }
        else add3(q, p);
    }
    static MA_DEV void dbl(Pt& p) {
        if constexpr (CV::A == 0) { dbl0(p); return; }
        else dbl3(p);
    }
    static MA_DEV void add3(const Pt& q, Pt& p) {
        int32_t B[10], T0[10], T1[10], T2[10], T3[10], T4[10];
        F::mul(p.X, q.X, T0);
        F::mul(p.Y, q.Y, T1);
        F::mul(p.Z, q.Z, T2);
        F::add(p.X, p.Y, T3);       // 8
        F::add(q.X, q.Y, T4);       // 8
        F::mul(T3, T4, T3);
        F::add(T0, T1, T4);         // 2
        F::sub(T3, T4, T3);         // 3
        F::add(p.Y, p.Z, T4);       // 8
        F::add(q.Y, q.Z, B);        // 8
        F::mul(T4, B, T4);
        F::add(T1, T2, B);          // 2
        F::sub(T4, B, T4);          // 3
        F::add(p.X, p.Z, p.X);      // 8
        F::add(q.Z, q.X, p.Y);      // 8
        F::mul(p.X, p.Y, p.X);
        F::add(T0, T2, p.Y);        // 2
        F::sub(p.X, p.Y, p.Y);      // 3
        mulb(T2, p.Z);
        F::sub(p.Y, p.Z, p.X);      // 4
        F::add(p.X, p.X, p.Z);      // 8
        F::add(p.X, p.Z, p.X);      // 12
        F::sub(T1, p.X, p.Z);       // 13
        F::add(p.X, T1, p.X);       // 13
        mulb(p.Y, p.Y);
        F::add(T2, T2, T1);         // 2
        F::add(T2, T1, T2);         // 3
        F::sub(p.Y, T2, p.Y);       // 4
        F::sub(p.Y, T0, p.Y);       // 5
        F::add(p.Y, p.Y, T1);       // 10
        F::add(p.Y, T1, p.Y);       // 15
        F::add(T0, T0, T1);         // 2
        F::add(T0, T1, T0);         // 3
        F::sub(T0, T2, T0);         // 6
        // the three outputs are sums of two products each; X3 and Z3 take one Montgomery reduction for both products
        // (weierstrass.c:158-174 reduces each product and adds), Y3 = 13 x 13 + 6 x 15 would overflow a column
        F::mul(T0, p.Y, T2);        // 6 x 15
        F::neg(p.Y, T1);            // 15
        F::mul(p.X, p.Z, p.Y);      // 13 x 13
        F::add(p.Y, T2, p.Y);       // 2
        F::mul2(p.X, T3, T4, T1, p.X);   // X3 = X T3 - T4 Y:  13 x 3 + 3 x 15
        F::mul2(p.Z, T4, T3, T0, p.Z);   // Z3 = Z T4 + T3 T0: 13 x 3 + 3 x 6
    }

    // P = 2P (RCB algorithm 6, a = -3; weierstrass.c:187-281)
    static MA_DEV void dbl3(Pt& p) {
        int32_t T0[10], T1[10], T2[10], T3[10], T4[10];
        F::sqr(p.X, T0);
        F::sqr(p.Y, T1);
        F::sqr(p.Z, T2);
        F::mul(p.X, p.Y, T3);
        F::mul(p.Y, p.Z, T4);
        F::add(T3, T3, T3);         // 2
        F::mul(p.Z, p.X, p.Z);
        F::add(p.Z, p.Z, p.Z);      // 2
        mulb(T2, p.Y);
        F::sub(p.Y, p.Z, p.Y);      // 3
        mulb(p.Z, p.Z);
        F::add(p.Y, p.Y, p.X);      // 6
        F::add(p.Y, p.X, p.Y);      // 9
        F::sub(T1, p.Y, p.X);       // 10
        F::add(p.Y, T1, p.Y);       // 10
        int32_t U[10];
        F::add(T2, T2, U);          // 2
        F::add(T2, U, T2);          // 3
        F::sub(p.Z, T2, p.Z);       // 4
        F::sub(p.Z, T0, p.Z);       // 5
        F::add(p.Z, p.Z, U);        // 10
        F::add(p.Z, U, p.Z);        // 15
        F::add(T0, T0, U);          // 2
        F::add(T0, U, T0);          // 3
        F::sub(T0, T2, T0);         // 6
        F::add(T4, T4, T4);         // 2
        // Y3 = Y X + T0 Z and X3 = X T3 - Z T4, each under one Montgomery reduction (weierstrass.c:249-275 reduces the four
        // products separately): 10 x 10 + 6 x 15 = 190 and 10 x 2 + 15 x 2
        F::mul2(p.Y, p.X, T0, p.Z, p.Y);
        F::neg(p.Z, U);             // 15
        F::mul2(p.X, T3, U, T4, p.X);
        F::mul(T4, T1, p.Z);
        F::add(p.Z, p.Z, p.Z);      // 2
        F::add(p.Z, p.Z, p.Z);      // 4
    }

    // P += (x2, y2), an AFFINE point of the curve (never the point at infinity): the complete mixed additions, RCB
    // algorithm 5 (a = -3) / 8 (a = 0), i.e. add() with Z2 = 1.  P may be anything, infinity included.  x2, y2 tight.
    static MA_DEV void madd(const int32_t* x2, const int32_t* y2, Pt& p) {
        if constexpr (CV::A == 0) {
            int32_t T0[10], T1[10], T2[10], T3[10], T4[10], U[10], V[10];
            F::mul(p.X, x2, T0);
            F::mul(p.Y, y2, T1);
            F::add(p.X, p.Y, T3);       // 2
            F::add(x2, y2, T4);         // 2
            F::mul(T3, T4, T3);
            F::add(T0, T1, T4);         // 2
            F::sub(T3, T4, T3);         // 3   X1 y2 + x2 Y1
            F::mul(y2, p.Z, T4);
            F::add(T4, p.Y, T4);        // 2   Y1 + y2 Z1
            F::mul(x2, p.Z, U);
            F::add(U, p.X, U);          // 2   X1 + x2 Z1
            F::add(T0, T0, V);          // 2
            F::add(T0, V, T0);          // 3   3 X1 x2
            CV::mul3b(p.Z, T2);         // 1   3b Z1
            CV::mul3b(U, U);            // 1
            F::add(T1, T2, V);          // 2
            F::sub(T1, T2, T1);         // 2
            F::neg(U, T2);              // 1
            F::mul2(T3, T1, T2, T4, p.X);    // 3 x 2 + 1 x 2
            F::mul2(U, T0, T1, V, p.Y);      // 1 x 3 + 2 x 2
            F::mul2(V, T4, T0, T3, p.Z);     // 2 x 2 + 3 x 3
        } else {
            int32_t B[10], T0[10], T1[10], T2[10], T3[10], T4[10];
            F::mul(p.X, x2, T0);
            F::mul(p.Y, y2, T1);
            F::add(p.X, p.Y, T3);       // 4
            F::add(x2, y2, T4);         // 2
            F::mul(T3, T4, T3);
            F::add(T0, T1, T4);         // 2
            F::sub(T3, T4, T3);         // 3   X1 y2 + x2 Y1
            F::mul(y2, p.Z, T4);
            F::add(T4, p.Y, T4);        // 3   Y1 + y2 Z1
            F::mul(x2, p.Z, B);
            F::add(B, p.X, p.Y);        // 3   X1 + x2 Z1
            F::copy(p.Z, T2);           // Z1 Z2 = Z1 (K = 1: Z of every sum is one fold of two products)
            mulb(T2, p.Z);
            F::sub(p.Y, p.Z, p.X);      // 4
            F::add(p.X, p.X, p.Z);      // 8
            F::add(p.X, p.Z, p.X);      // 12
            F::sub(T1, p.X, p.Z);       // 13
            F::add(p.X, T1, p.X);       // 13
            mulb(p.Y, p.Y);
            F::add(T2, T2, T1);         // 2
            F::add(T2, T1, T2);         // 3
            F::sub(p.Y, T2, p.Y);       // 4
            F::sub(p.Y, T0, p.Y);       // 5
            F::add(p.Y, p.Y, T1);       // 10
            F::add(p.Y, T1, p.Y);       // 15
            F::add(T0, T0, T1);         // 2
            F::add(T0, T1, T0);         // 3
            F::sub(T0, T2, T0);         // 6
            F::mul(T0, p.Y, T2);        // 6 x 15
            F::neg(p.Y, T1);            // 15
            F::mul(p.X, p.Z, p.Y);      // 13 x 13
            F::add(p.Y, T2, p.Y);       // 2
            F::mul2(p.X, T3, T4, T1, p.X);   // 13 x 3 + 3 x 15
            F::mul2(p.Z, T4, T3, T0, p.Z);   // 13 x 3 + 3 x 6
        }
    }

    static MA_DEV void load_point(const spint* X, const spint* Y, const spint* Z, Pt& p) {
        F::from52(X, p.X);
        F::from52(Y, p.Y);
        F::from52(Z, p.Z);
    }
    static MA_DEV void put(uint64_t* tab, size_t tstride, int entry, const Pt& p) {
        uint64_t w[5];
        F::pack(p.X, w);
        static_for<0, 5>([&](auto K) { tab[(size_t)(entry * 15 + K) * tstride] = w[K]; });
        F::pack(p.Y, w);
        static_for<0, 5>([&](auto K) { tab[(size_t)(entry * 15 + 5 + K) * tstride] = w[K]; });
        F::pack(p.Z, w);
        static_for<0, 5>([&](auto K) { tab[(size_t)(entry * 15 + 10 + K) * tstride] = w[K]; });
    }
    static MA_DEV void get(const uint64_t* tab, size_t tstride, int entry, Pt& p) {
        uint64_t w[15];
        static_for<0, 15>([&](auto K) { w[K] = tab[(size_t)(entry * 15 + K) * tstride]; });
        F::unpack(w, p.X);
        F::unpack(w + 5, p.Y);
        F::unpack(w + 10, p.Z);
    }
    // entries base .. base + COUNT - 1 = P, 2P, ..., COUNT P: even multiples by doubling, odd ones as (k-1)P + P, rolled
    // into one loop so that the instruction stream holds a single copy of dbl and add for the table
    // (P itself comes back from entry 0 for every odd multiple: held in registers across the loop it cost 14-28 spilled registers at three
    // waves per SIMD)
    template <int COUNT>
    static MA_DEV void build_table(const Pt& p, uint64_t* tab, size_t tstride, int base) {
        put(tab, tstride, base, p);
#pragma unroll 1
        for (int k = 2; k <= COUNT; k++) {
            Pt t;
            get(tab, tstride, base + ((k & 1) ? k - 2 : (k >> 1) - 1), t);
            if (k & 1) {
                Pt q;
                get(tab, tstride, base, q);
                add(q, t);
            } else {
                dbl(t);
            }
            put(tab, tstride, base + k - 1, t);
        }
    }
    // sign * table[base + m - 1] (m = 0: the point at infinity), every entry read; COUNT entries scanned two per round
    template <int COUNT>
    static MA_DEV void lookup(const uint64_t* tab, size_t tstride, int base, uint32_t m, bool neg, Pt& q) {
        uint64_t sel[15];
        {
            int32_t o[10];
            uint64_t w[5];
            F::set_one(o);
            F::pack(o, w);
            static_for<0, 15>([&](auto K) { sel[K] = (K >= 5 && K < 10) ? w[K - 5] : 0u; });
        }
        // the memory clobbers keep the (loop-invariant) table loads inside the iteration (see ed28.h)
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" ::: "memory");
#endif
#pragma unroll 1
        for (int e = 0; e < COUNT; e += 2) {
            uint64_t ent[2][15];
            static_for<0, 2>([&](auto EI) {
                static_for<0, 15>([&](auto K) { ent[EI][K] = tab[(size_t)((base + e + EI) * 15 + K) * tstride]; });
            });
            static_for<0, 2>([&](auto EI) {
                const bool hit = (m == (uint32_t)(e + EI + 1));
                static_for<0, 15>([&](auto K) {
                    const uint64_t a = ent[EI][K], b = sel[K];
                    sel[K] = hit ? a : b;
                });
            });
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" ::: "memory");
#endif
        }
        int32_t ny[10];
        F::unpack(sel, q.X);
        F::unpack(sel + 5, q.Y);
        F::unpack(sel + 10, q.Z);
        F::neg(q.Y, ny);
        F::select(neg, q.Y, ny, q.Y);
    }
    // ecnXXXget of a projective point: canonical affine words; Z = 0 leaves as (0, 1) (weierstrass.c:299-310)
    static MA_DEV void affine_words(const Pt& p, uint64_t* xw, uint64_t* yw) {
        int32_t zi[10], ax[10], ay[10];
        uint64_t zw[4];
        F::to_words(p.Z, zw);
        const bool z0 = (zw[0] | zw[1] | zw[2] | zw[3]) == 0;
        F::invert(p.Z, zi);
        F::mul(p.X, zi, ax);
        F::mul(p.Y, zi, ay);
        F::to_words(ax, xw);
        F::to_words(ay, yw);
        const uint64_t keep = lane_mask(!z0);             // (field.h: a mask, not a select -- no EXEC region around the export)
        static_for<0, 4>([&](auto K) {
            xw[K] &= keep;
            yw[K] = (yw[K] & keep) | (K == 0 ? (1u & ~keep) : 0u);
        });
    }
    // affine_words for G points with ONE inversion (Montgomery's trick on the prefix products of the Z); a point at infinity
    // takes Z = 1 into the shared product and leaves as (0, 1) as above.  xw, yw: G x 4 words.
    template <int G>
    static MA_DEV void affine_words_many(const Pt* pts, uint64_t (*xw)[4], uint64_t (*yw)[4]) {
        int32_t z[G][10], pre[G][10], inv[10], t[10], u[10], one[10];
        bool inf_[G];
        F::set_one(one);
        static_for<0, G>([&](auto GI) {
            constexpr int g = GI;
            uint64_t zw[4];
            F::to_words(pts[g].Z, zw);
            inf_[g] = (zw[0] | zw[1] | zw[2] | zw[3]) == 0;
            F::select(inf_[g], pts[g].Z, one, z[g]);
            if constexpr (g == 0) F::copy(z[0], pre[0]);
            else F::mul(pre[g - 1], z[g], pre[g]);
        });
        F::invert(pre[G - 1], inv);
        static_for<0, G>([&](auto GI) {
            constexpr int g = G - 1 - GI;
            if constexpr (g > 0) {
                F::mul(inv, pre[g - 1], t);         // 1 / z_g
                F::mul(inv, z[g], inv);             // 1 / (z_0 ... z_{g-1})
            } else {
                F::copy(inv, t);
            }
            F::mul(pts[g].X, t, u);
            F::to_words(u, xw[g]);
            F::mul(pts[g].Y, t, u);
            F::to_words(u, yw[g]);
            const uint64_t keep = lane_mask(!inf_[g]);
            static_for<0, 4>([&](auto K) {
                xw[g][K] &= keep;
                yw[g][K] = (yw[g][K] & keep) | (K == 0 ? (1u & ~keep) : 0u);
            });
        });
    }
};

// s = e + bias (bias: a 1 in bit positions b < BITS with b % W == W - 1), then left-aligned so that the top window
// (bits BITS-W .. BITS-1 of s) is the top of w[4]
template <int W, int BITS>
MA_DEV void wn26_recode(const uint64_t* ew, uint64_t* w) {
    constexpr auto cw = [](int k) {
        uint64_t v = 0;
        for (int b = 0; b < 64; b++) {
            const int pos = 64 * k + b;
            if (pos < BITS && pos % W == W - 1) v |= (uint64_t)1 << b;
        }
        return v;
    };
    unsigned __int128 acc = 0;
    uint64_t s[5];
    static_for<0, 5>([&](auto K) {
        constexpr int k = K;
        acc += (unsigned __int128)(k < 4 ? ew[k < 4 ? k : 0] : 0) + cw(k);
        s[k] = (uint64_t)acc;
        acc >>= 64;
    });
    constexpr int SH = 320 - BITS;           // 60 for 260 bits, 62 for 258
    static_for<0, 5>([&](auto KK) {
        constexpr int k = 4 - KK;
        w[k] = s[k] << SH;
        if constexpr (k > 0) w[k] |= s[k - 1] >> (64 - SH);
    });
}
template <int W>
MA_DEV uint32_t wn26_take(uint64_t* w) {
    const uint32_t win = (uint32_t)(w[4] >> (64 - W));
    static_for<0, 5>([&](auto KK) {
        constexpr int k = 4 - KK;
        w[k] <<= W;
        if constexpr (k > 0) w[k] |= w[k - 1] >> (64 - W);
    });
    return win;
}

// ---- digit sources and table accessors (round 4; the concepts of ed28.h): Regs = the shift registers above, Lds = one byte per
// window in the lane's column of an LDS array, written before the point is loaded; the table accessor is ed28.h's TabStrided
// (host check, round-3 layout) or TabSlab (per-wave slab, row addresses formed at the access).  window(i) for i = 0, 1, ... in order.
template <int W, int BITS>
struct WnRegs {
    uint64_t w[5];
    uint32_t flag = 0;
    MA_DEV void init(const uint64_t* ew) { wn26_recode<W, BITS>(ew, w); }
    MA_DEV uint32_t window(int) { return wn26_take<W>(w); }
    MA_DEV void park(uint32_t v) { flag = v; }                  // one per-lane flag kept beside the digits (WnLds: in LDS, not in a register)
    MA_DEV uint32_t parked() const { return flag; }
};
template <int W, int BITS>
struct WnLds {
    static constexpr int COUNT = (BITS + W - 1) / W;            // 65 windows of 4 bits (260), 86 of 3 bits (258)
    static constexpr int ROWS = COUNT + 1;                      // + one row for a per-lane flag (park / parked)
    unsigned char* col;
    static MA_DEV void fill(const uint64_t* ew, unsigned char* col) {
        WnRegs<W, BITS> r;
        r.init(ew);
#pragma unroll 1
        for (int i = 0; i < COUNT; i++) col[(size_t)i * 64] = (unsigned char)r.window(i);
    }
    MA_DEV uint32_t window(int i) const { return col[(size_t)i * 64]; }
    MA_DEV void park(uint32_t v) { col[(size_t)COUNT * 64] = (unsigned char)v; }
    MA_DEV uint32_t parked() const { return col[(size_t)COUNT * 64]; }
};
struct WnTabStrided {
    uint64_t* tab;
    size_t tstride;
    MA_DEV uint64_t* origin() const { return tab; }
    MA_DEV size_t stride() const { return tstride; }
};
struct WnTabSlab {
    uint64_t* base;                         // the wave's slab [word][64 lanes], wave-uniform
    unsigned lane;
    MA_DEV uint64_t* origin() const {
        unsigned l = lane;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(l));
#endif
        return base + l;
    }
    static MA_DEV constexpr size_t stride() { return 64; }
};

// One fused P-256 scalar multiplication + affine export.
//   ew: the scalar as four little-endian 64-bit words (the caller has byte-swapped the big-endian record);
//   X, Y, Z: the projective point, 5 x 52-bit limbs each (field.c form); tab: this lane's table slot (NIST256_TABLE_WORDS
//   words, tstride apart); xw, yw: canonical affine coordinates, four little-endian words each.
template <class CV, class TAB, class DIG>
MA_DEV void wn26_mul_acc(DIG& dig, const spint* X, const spint* Y, const spint* Z, const TAB& T,
                         typename Wn26<CV>::Pt& R) {
    using E = Wn26<CV>;
    typename E::Pt Q;
    E::load_point(X, Y, Z, Q);
    E::template build_table<8>(Q, T.origin(), T.stride(), 0);
    E::inf(R);
#pragma unroll 1
    for (int i = 0; i < 65; i++) {
        const int dgt = (int)dig.window(i) - 8;                 // [-8, 7]
        const bool neg = dgt < 0;
        const uint32_t m = (uint32_t)(neg ? -dgt : dgt);        // 0..8
        if (i != 0) {
#pragma unroll 1
            for (int j = 0; j < 4; j++) E::dbl(R);
        }
        E::template lookup<8>(T.origin(), T.stride(), 0, m, neg, Q);
        E::add(Q, R);
    }
}
template <class CV>
MA_DEV void wn26_mul_acc(const uint64_t* ew, const spint* X, const spint* Y, const spint* Z, uint64_t* tab, size_t tstride,
                         typename Wn26<CV>::Pt& R) {
    WnRegs<4, 260> dig;
    dig.init(ew);
    wn26_mul_acc<CV>(dig, X, Y, Z, WnTabStrided{tab, tstride}, R);
}
template <class CV, class TAB, class DIG>
MA_DEV void wn26_mul_get_dig(DIG& dig, const spint* X, const spint* Y, const spint* Z, const TAB& T, uint64_t* xw, uint64_t* yw) {
    typename Wn26<CV>::Pt R;
    wn26_mul_acc<CV>(dig, X, Y, Z, T, R);
    Wn26<CV>::affine_words(R, xw, yw);
}
template <class CV>
MA_DEV void wn26_mul_get_one(const uint64_t* ew, const spint* X, const spint* Y, const spint* Z, uint64_t* tab, size_t tstride,
                             uint64_t* xw, uint64_t* yw) {
    typename Wn26<CV>::Pt R;
    wn26_mul_acc<CV>(ew, X, Y, Z, tab, tstride, R);
    Wn26<CV>::affine_words(R, xw, yw);
}

// Fused double multiplication + affine export: the affine coordinates of e*P + f*Q (ecnXXXmul2 followed by ecnXXXget,
// the verification pattern nist256.c:251-256).  Signed 3-bit digits (e' = e + sum_{i<86} 4*8^i, digit = window - 4 in
// [-4, 3]) so that the two tables {1..4}P and {1..4}Q share the eight entry slots of the same per-lane workspace; per
// window three doublings and two additions, all lookups scan their table.  (The reference's mul2 is a joint sparse form
// with data-dependent branches; any evaluation reaches the same affine point.)  An infinite result leaves as (0, 1).
template <class CV, class TAB, class DIG>
MA_DEV void wn26_mul2_get_dig(DIG& dige, const spint* PX, const spint* PY, const spint* PZ,
                              DIG& digf, const spint* QX, const spint* QY, const spint* QZ,
                              const TAB& T, uint64_t* xw, uint64_t* yw) {
    using E = Wn26<CV>;
    typename E::Pt R, Q;
    E::load_point(PX, PY, PZ, Q);
    E::template build_table<4>(Q, T.origin(), T.stride(), 0);
    E::load_point(QX, QY, QZ, Q);
    E::template build_table<4>(Q, T.origin(), T.stride(), 4);
    E::inf(R);
#pragma unroll 1
    for (int i = 0; i < 86; i++) {
        const int de = (int)dige.window(i) - 4, df = (int)digf.window(i) - 4;       // [-4, 3]
        if (i != 0) {
#pragma unroll 1
            for (int j = 0; j < 3; j++) E::dbl(R);
        }
#pragma unroll 1
        for (int which = 0; which < 2; which++) {
            const int dgt = which ? df : de;
            const bool neg = dgt < 0;
            const uint32_t m = (uint32_t)(neg ? -dgt : dgt);    // 0..4
            E::template lookup<4>(T.origin(), T.stride(), 4 * which, m, neg, Q);
            E::add(Q, R);
        }
    }
    E::affine_words(R, xw, yw);
}
template <class CV>
MA_DEV void wn26_mul2_get_one(const uint64_t* ew, const spint* PX, const spint* PY, const spint* PZ,
                              const uint64_t* fw, const spint* QX, const spint* QY, const spint* QZ,
                              uint64_t* tab, size_t tstride, uint64_t* xw, uint64_t* yw) {
    WnRegs<3, 258> de, df;
    de.init(ew);
    df.init(fw);
    wn26_mul2_get_dig<CV>(de, PX, PY, PZ, df, QX, QY, QZ, WnTabStrided{tab, tstride}, xw, yw);
}

// Fused GENERATOR multiplication + affine export: the affine coordinates of e*G -- ecnXXXgen, ecnXXXmul, ecnXXXget, the
// opening of key generation and signing in the reference's ECDSA code (nist256.c:150-161 NIST256_KEY_PAIR, 214-222
// NIST256_SIGN).  With the base point fixed the doublings disappear: e' = e + sum_i 2^(W-1) 2^(W i), digit_i = window_i(e')
// - 2^(W-1), and e*G = sum_i digit_i * (2^(W i) G) with the NW x 2^(W-1) affine multiples precomputed (generated/comb_<C>.h:
// W = 5, 52 windows x 16 entries, 66 560 bytes, the same table for every lane; W = 4 is 6 % slower, W = 6 -- 43 mixed
// additions but 32 entries to scan -- no faster than W = 5 on P-256 and slower on secp256k1).  Per window ALL its entries are read through wave-uniform addresses (TAB: the
// table in the constant address space, scalar loads) and selected by lane predication, the sign negates y, and one
// complete MIXED addition follows; a zero digit adds (0, 0) and keeps the old sum.  52 mixed additions + a share of one
// inversion per scalar against 256 doublings + 65 additions: the same bytes as ecn gen + ecn mul + ecn get for every scalar.
// the walk over the windows of e: step(sx, sy, zero) is handed +-(the window's selected multiple of G), affine, or zero = true for a zero digit
template <class CV, class TAB, class STEP>
MA_DEV void wn26_mulgen_walk(const uint64_t* ew, STEP step) {
    using F = typename CV::F;
    constexpr int W = TAB::W, NW = TAB::NW, E2 = 1 << (W - 1);       // window width, windows, entries per window
    static_assert(W * NW >= 257 && W * NW <= 316, "e + bias must fit the windows and five words");
    uint64_t w[5];
    {
        uint64_t t[5];
        wn26_recode<W, W * NW>(ew, t);              // left-aligned: undo, the windows are taken from the bottom here
        constexpr int SH = 320 - W * NW;
        static_for<0, 5>([&](auto K) {
            constexpr int k = K;
            w[k] = t[k] >> SH;
            if constexpr (k < 4) w[k] |= t[k + 1] << (64 - SH);
        });
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
        // selection as OR of masked entries (exactly one mask is set, none for a zero digit): the entries sit in scalar
        // registers, and v_and_or_b32 takes one of those next to two vector operands -- a v_cndmask would need a move first
        // (its lane mask already uses the one scalar operand an instruction may read)
        int32_t sx[10], sy[10], ny[10];
        static_for<0, 10>([&](auto K) { sx[K] = 0; sy[K] = 0; });
        static_for<0, E2>([&](auto MM) {
            constexpr int mm = MM;
            uint32_t mask = (m == (uint32_t)(mm + 1)) ? 0xffffffffu : 0u;
#if defined(__HIP_DEVICE_COMPILE__)
            asm("" : "+v"(mask));       // opaque: otherwise the compiler turns (entry & mask) back into a select with a move
#endif
            static_for<0, 10>([&](auto K) {
                sx[K] = (int32_t)((uint32_t)sx[K] | ((uint32_t)TAB::get(((i * E2 + mm) * 2 + 0) * 10 + K) & mask));
                sy[K] = (int32_t)((uint32_t)sy[K] | ((uint32_t)TAB::get(((i * E2 + mm) * 2 + 1) * 10 + K) & mask));
            });
        });
        F::neg(sy, ny);
        F::select(neg, sy, ny, sy);
        step(sx, sy, m == 0);
    }
}
template <class CV, class TAB, bool INIT = true>       // INIT = false: R += e*G (R holds a sum already)
MA_DEV void wn26_mulgen_acc(const uint64_t* ew, typename Wn26<CV>::Pt& R) {
    using E = Wn26<CV>;
    using F = typename CV::F;
    if constexpr (INIT) E::inf(R);
    wn26_mulgen_walk<CV, TAB>(ew, [&](const int32_t* sx, const int32_t* sy, bool keep) {
        typename E::Pt S = R;
        E::madd(sx, sy, S);
        F::select(keep, S.X, R.X, R.X);
        F::select(keep, S.Y, R.Y, R.Y);
        F::select(keep, S.Z, R.Z, R.Z);
    });
}
template <class CV, class TAB>
MA_DEV void wn26_mulgen_get_one(const uint64_t* ew, uint64_t* xw, uint64_t* yw) {
    typename Wn26<CV>::Pt R;
    wn26_mulgen_acc<CV, TAB>(ew, R);
    Wn26<CV>::affine_words(R, xw, yw);
}
// G scalars per lane, one inversion for all of them (the inversion is a fifth of the single-scalar kernel).  load(g, ew)
// fetches the g-th scalar of this lane; the window loop is rolled over g (one copy in the instruction stream).
template <class CV, class TAB, int G, class LOAD>
MA_DEV void wn26_mulgen_get_many(LOAD load, uint64_t (*xw)[4], uint64_t (*yw)[4]) {
    typename Wn26<CV>::Pt Rs[G], R;
#pragma unroll 1
    for (int g = 0; g < G; g++) {
        uint64_t ew[4];
        load(g, ew);
        wn26_mulgen_acc<CV, TAB>(ew, R);
        static_for<0, G>([&](auto GI) { if (g == GI) Rs[GI] = R; });       // (static register indices only)
    }
    Wn26<CV>::template affine_words_many<G>(Rs, xw, yw);
}

// Fused e*G + f*Q + affine export: the verification pattern ecnXXXmul2(u, &G, v, &Q, &Q); ecnXXXget (nist256.c:251-256) --
// the first point of the reference's double multiplication there is always the GENERATOR.  f*Q runs as in
// wn26_mul_get_one (4-bit windows on a per-lane table of Q), e*G joins through the fixed-base table with mixed additions
// and no doublings of its own (wn26_mulgen_acc): 256 doublings + 65 additions + 52 mixed additions against the 258 + 172 of
// the general wn26_mul2_get_one.  Same bytes as ecn gen, ecn mul2, ecn get.
template <class CV, class TAB>
MA_DEV void wn26_mulgen2_get_one(const uint64_t* ew, const uint64_t* fw, const spint* QX, const spint* QY, const spint* QZ,
                                 uint64_t* tab, size_t tstride, uint64_t* xw, uint64_t* yw) {
    typename Wn26<CV>::Pt R;
    wn26_mul_acc<CV>(fw, QX, QY, QZ, tab, tstride, R);
    wn26_mulgen_acc<CV, TAB, false>(ew, R);
    Wn26<CV>::affine_words(R, xw, yw);
}
template <class CV, class COMB, class TAB, class DIG>
MA_DEV void wn26_mulgen2_get_dig(const uint64_t* ew, DIG& digf, const spint* QX, const spint* QY, const spint* QZ, const TAB& T,
                                 uint64_t* xw, uint64_t* yw) {
    typename Wn26<CV>::Pt R;
    wn26_mul_acc<CV>(digf, QX, QY, QZ, T, R);
    wn26_mulgen_acc<CV, COMB, false>(ew, R);
    Wn26<CV>::affine_words(R, xw, yw);
}

// the P-256 entry points (C: generated/curve_NIST256.h, documentation only: the constants of this form are in CvNist256)
template <class C>
MA_DEV void nist256_mul_get_one(const uint64_t* ew, const spint* X, const spint* Y, const spint* Z, uint64_t* tab, size_t tstride,
                                uint64_t* xw, uint64_t* yw) {
    wn26_mul_get_one<CvNist256>(ew, X, Y, Z, tab, tstride, xw, yw);
}
template <class C>
MA_DEV void nist256_mul2_get_one(const uint64_t* ew, const spint* PX, const spint* PY, const spint* PZ,
                                 const uint64_t* fw, const spint* QX, const spint* QY, const spint* QZ,
                                 uint64_t* tab, size_t tstride, uint64_t* xw, uint64_t* yw) {
    wn26_mul2_get_one<CvNist256>(ew, PX, PY, PZ, fw, QX, QY, QZ, tab, tstride, xw, yw);
}

}  // namespace ma
