// modarith_amd/csrc/wj26.h -- the fused P-256 scalar multiplication k P in JACOBIAN coordinates (round 5).
//
// wn26.h runs the complete projective formulas of Renes-Costello-Batina the reference uses (weierstrass.c:68-281): for a = -3 a
// doubling is 8M + 3S + 2 multiplications by b, 1 649 multiply-adds on fm26.h.  The fused entry points only let canonical affine
// bytes out (ecnXXXmul followed by ecnXXXget, nist256.c:155-161, 219-222, 251-256), so the coordinate system inside is as free as
// the window width and the limb form: in Jacobian coordinates (x = X / Z^2, y = Y / Z^3) the a = -3 doubling is 3M + 5S = 927
// multiply-adds (Bernstein-Lange dbl-2001-b); the general addition 11M + 5S with one shared reduction = 2 035 (add-2007-bl) builds the
// window table, whose entries a shared inversion then brings to Z = 1 (wn_affine.h), so that the window loop adds with the MIXED
// formula, 7M + 4S = 1 360 (madd-2007-bl).  256 doublings + 65 additions: 361 k multiply-adds instead of 547 k.
//
// The Jacobian addition is NOT complete: it fails for R = +-Q and for either operand at infinity.  Where those can happen is decided
// by the SCALAR alone, because the group has prime order n and cofactor 1 (every point of the curve other than infinity has order n):
//   * the scalar is reduced mod n first (e < 2^256 < 2n: one masked subtraction; e P = (e mod n) P for every point of the curve);
//   * with the signed digits of wn26.h (e' = e + sum 8 * 16^i, d_j = window_j(e') - 8), the accumulator before digit j is added is
//     16 m P with m = floor(e' / 16^(j+1)) - 0x88..8 >= 0, and 16 m + d_j = floor(e' / 16^j) - 0x88..8 <= e / 16^j + 1.  For j >= 1 both
//     16 m and 16 m + d_j are below n / 16 + 17 < n - 8: 16 m = +-d_j (mod n) forces m = 0 = d_j.  So for every digit but the LAST the
//     only exceptional cases are an accumulator at infinity (m = 0: a lane flag, the sum is then the table entry), a zero digit or
//     an input point at infinity (the sum is the accumulator), and doubling never fails (a point of odd order has Y != 0);
//   * for the last digit the accumulator CAN meet +-Q (e = n - 2: 16 m = n - 1, d_0 = -1), so the last addition is the complete one of
//     wn26.h (RCB algorithm 4) on the two points converted to homogeneous coordinates (X Z : Y : Z^3) -- 2M + 1S each -- and the
//     result leaves in those, ready for wn26.h's export and for the complete mixed additions of the generator part (e G + f Q, where
//     Q = k G for an unknown k and the accumulator can meet any table entry).
//   * the table {1..8}P: 2P, 4P, 6P, 8P by doubling, 3P, 5P, 7P as (k-1)P + P -- no two multiples below n coincide.
// Input points off the curve mean nothing on either side (wn26.h).  Flags are lane masks: the same instruction and address sequence
// for every scalar and point.  The double multiplication e P + f Q of two caller points (mul2_acc_aff below) has no such argument --
// its accumulator depends on both points and can meet a table entry anywhere in the loop: it tests every addition for that and redoes
// it with the complete formula, wave by wave (variable time, public inputs).
//
// K (fm26.h: |limb| <= K 2^26; a product needs K_f K_g <= 190, every limb below 2^31) is given in the comments.
#pragma once
#include "wn26.h"
#include "wn_affine.h"

namespace ma {

struct Wj26 {
    using F = Fm26;
    using E = Wn26<CvNist256>;
    using Pt = E::Pt;                       // (X, Y, Z), Jacobian here

    // the group order, little-endian words
    static constexpr uint64_t n_(int i) { constexpr uint64_t v[4] = {0xF3B9CAC2FC632551ull, 0xBCE6FAADA7179E84ull, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFF00000000ull}; return v[i]; }
    static MA_DEV void reduce_scalar(const uint64_t* ew, uint64_t* k) {
        uint64_t d[4], bw = 0;
        static_for<0, 4>([&](auto I) {
            constexpr uint64_t nI = n_(I);
            const uint64_t x = ew[I] - nI, b1 = ew[I] < nI;
            d[I] = x - bw;
            bw = b1 | ((x < bw) ? 1u : 0u);
        });
        const uint64_t keep = (uint64_t)0 - bw;                 // borrow: e < n
        static_for<0, 4>([&](auto I) { k[I] = (ew[I] & keep) | (d[I] & ~keep); });
    }

    // P = 2P (dbl-2001-b).  In: X, Y K <= 9, Z K <= 3.  Out: X K = 9, Y K = 1, Z K = 3.  Z = 0 stays Z = 0.
    static MA_DEV void dbl(Pt& p) {
        int32_t d[10], g[10], b[10], a[10], t0[10], t1[10];
        F::sqr(p.Z, d);             // delta
        F::sqr(p.Y, g);             // gamma
        F::mul(p.X, g, b);          // beta
        F::sub(p.X, d, t0);         // 10
        F::add(p.X, d, t1);         // 10
        F::mul(t0, t1, a);
        F::add(a, a, t0);
        F::add(a, t0, a);           // 3   alpha = 3 (X - delta)(X + delta)
        F::add(p.Y, p.Z, t0);       // 12
        F::sqr(t0, t0);
        F::sub(t0, g, t0);
        F::sub(t0, d, p.Z);         // 3   Z3 = (Y + Z)^2 - gamma - delta
        F::add(b, b, b);
        F::add(b, b, b);            // 4   4 beta
        F::sqr(a, t0);
        F::sub(t0, b, t0);
        F::sub(t0, b, p.X);         // 9   X3 = alpha^2 - 8 beta
        F::sub(b, p.X, t1);         // 13
        F::add(g, g, t0);
        F::add(t0, t0, t0);
        F::add(t0, t0, t0);
        F::neg(t0, t0);             // 8   -8 gamma
        F::mul2(a, t1, t0, g, p.Y); //     Y3 = alpha (4 beta - X3) - 8 gamma^2 under one reduction: 3 x 13 + 8 x 1 (the price of mul + sqr)
    }
    // P += Q (add-2007-bl), neither at infinity, P != +-Q.  In: K <= 9 (X, Y), <= 3 (Z) on both.  Out: X K = 4, Y, Z K = 1.
    static MA_DEV void add(const Pt& q, Pt& p) {
        int32_t z1z1[10], z2z2[10], u1[10], u2[10], s1[10], s2[10], h[10], i_[10], j[10], r[10], v[10], t[10];
        F::sqr(p.Z, z1z1);
        F::sqr(q.Z, z2z2);
        F::mul(p.X, z2z2, u1);
        F::mul(q.X, z1z1, u2);
        F::mul(q.Z, z2z2, t);
        F::mul(p.Y, t, s1);
        F::mul(p.Z, z1z1, t);
        F::mul(q.Y, t, s2);
        F::sub(u2, u1, h);          // 2
        F::add(h, h, t);            // 4
        F::sqr(t, i_);              //     I = (2H)^2
        F::mul(h, i_, j);           //     J = H I
        F::sub(s2, s1, r);
        F::add(r, r, r);            // 4   r = 2 (S2 - S1)
        F::mul(u1, i_, v);          //     V = U1 I
        F::add(p.Z, q.Z, t);        // 6
        F::sqr(t, t);
        F::sub(t, z1z1, t);
        F::sub(t, z2z2, t);         // 3
        F::mul(t, h, p.Z);          //     Z3 = ((Z1 + Z2)^2 - Z1Z1 - Z2Z2) H
        F::sqr(r, t);
        F::sub(t, j, t);
        F::sub(t, v, t);
        F::sub(t, v, p.X);          // 4   X3 = r^2 - J - 2V
        F::sub(v, p.X, t);          // 5
        F::add(s1, s1, s1);         // 2
        F::neg(s1, s1);
        F::mul2(r, t, s1, j, p.Y);  //     Y3 = r (V - X3) - 2 S1 J:  4 x 5 + 2 x 1
    }
    // homogeneous (X : Y : Z) of the reference's form -> Jacobian (X Z, Y Z^2, Z); Z = 0 gives (0, 0, 0)
    static MA_DEV void from_projective(Pt& p) {
        int32_t zz[10];
        F::sqr(p.Z, zz);
        F::mul(p.X, p.Z, p.X);
        F::mul(p.Y, zz, p.Y);
    }
    // Jacobian -> homogeneous (X Z : Y : Z^3), K = 1 throughout; inf: the point is at infinity whatever the coordinates say -> (0 : 1 : 0)
    static MA_DEV void to_projective(bool inf, Pt& p) {
        int32_t zz[10], o[10], z[10];
        F::set_one(o);
        F::zero(z);
        F::sqr(p.Z, zz);
        F::mul(p.X, p.Z, p.X);
        F::mul(p.Y, o, p.Y);        // Y as it is, limbs carried (the complete addition takes K <= 4)
        F::mul(p.Z, zz, p.Z);
        F::select(inf, p.X, z, p.X);
        F::select(inf, p.Y, o, p.Y);
        F::select(inf, p.Z, z, p.Z);
    }
    static MA_DEV bool is_zero(const int32_t* f) {
        uint64_t w[4];
        F::to_words(f, w);          // canonical
        return (w[0] | w[1] | w[2] | w[3]) == 0;
    }

    // entries base .. base + 7 = P, 2P, ..., 8P (Jacobian), one copy of dbl and add in the instruction stream (wn26.h build_table)
    static MA_DEV void build_table(const Pt& p, uint64_t* tab, size_t tstride, int base = 0) {
        E::put(tab, tstride, base, p);
#pragma unroll 1
        for (int k = 2; k <= 8; k++) {
            Pt t;
            E::get(tab, tstride, base + ((k & 1) ? k - 2 : (k >> 1) - 1), t);
            if (k & 1) {
                Pt q;
                E::get(tab, tstride, base, q);          // (P from entry 0, not held across the loop: wn26.h build_table)
                add(q, t);
            } else {
                dbl(t);
            }
            E::put(tab, tstride, base + k - 1, t);
        }
    }

    // ---- k P on an AFFINE table (wn_affine.h): the multiples of P are computed by table_of() in a kernel of their own and brought to
    // Z = 1 by k_wn_table_affine; the window loop adds them with madd() below.  R = (the scalar whose digits dig delivers) * P,
    // homogeneous on return; the digits must be those of a scalar below n (reduce_scalar before the recoding).  Flags for an
    // accumulator / digit at infinity, the last addition the complete MIXED one of wn26.h -- the header.
    template <class LD>
    static MA_DEV void table_of(LD load, const WnAffWs& ws, size_t t, int base = 0) {
        Pt Q;
        {
            spint X[5], Y[5], Z[5];
            load(X, Y, Z);
            E::load_point(X, Y, Z, Q);
        }
        from_projective(Q);                                         // Z = 0 stays Z = 0 in every multiple: k_wn_table_affine flags the record
        build_table(Q, ws.T + t, ws.m, base);
    }
    template <class DIG>
    static MA_DEV void mul_acc_aff(DIG& dig, const WnAffWs& ws, size_t t, Pt& R) {
        dig.park(ws.flag[t]);
        F::set_one(R.X);
        F::set_one(R.Y);
        F::zero(R.Z);
        bool rinf = true;
        int32_t sx[10], sy[10];
#pragma unroll 1
        for (int i = 0; i < 64; i++) {
            const int dgt = (int)dig.window(i) - 8;                 // [-8, 7]
            const bool neg = dgt < 0;
            const uint32_t m = (uint32_t)(neg ? -dgt : dgt);        // 0..8
            if (i != 0) {
#pragma unroll 1
                for (int j = 0; j < 4; j++) dbl(R);
            }
            wn_affine_lookup<F>(ws, t, m, neg, sx, sy);
            const uint32_t pk = dig.parked();
            const bool qinf = ((m == 0) | (pk != 0)) != 0;
            Pt S = R;
            madd(sx, sy, S);
            int32_t one[10], u[10];
            F::set_one(one);
            F::select(rinf, S.X, sx, u);
            F::select(qinf, u, R.X, R.X);
            F::select(rinf, S.Y, sy, u);
            F::select(qinf, u, R.Y, R.Y);
            F::select(rinf, S.Z, one, u);
            F::select(qinf, u, R.Z, R.Z);
            rinf = rinf && qinf;
        }
        {
            const int dgt = (int)dig.window(64) - 8;
            const bool neg = dgt < 0;
            const uint32_t m = (uint32_t)(neg ? -dgt : dgt);
#pragma unroll 1
            for (int j = 0; j < 4; j++) dbl(R);
            wn_affine_lookup<F>(ws, t, m, neg, sx, sy);
            const uint32_t pk = dig.parked();
            const bool qinf = ((m == 0) | (pk != 0)) != 0;
            to_projective(rinf, R);
            Pt S = R;
            E::madd(sx, sy, S);                                     // complete: R may be anything, the affine point is finite when it is used
            F::select(qinf, S.X, R.X, R.X);
            F::select(qinf, S.Y, R.Y, R.Y);
            F::select(qinf, S.Z, R.Z, R.Z);
        }
    }

    // ---- e G through the fixed-base table (wn26.h wn26_mulgen_acc) with the Jacobian MIXED addition (madd-2007-bl, 7M + 4S: 1 360
    // multiply-adds against the 1 740 of the complete mixed addition).  The window digits d_i = window_i(e + sum 16 * 32^i) - 16 are taken
    // from the bottom; before window i the accumulator is s G with |s| < 0.52 * 32^i (the low windows' value) and the table point is
    // d_i 32^i G with |d_i 32^i| >= 32^i: never +-s as integers, and |s| + |d_i| 32^i < n for i <= 50.  At i = 51 (32^51 = 2^255, d in
    // {0, 1, 2} for e < n) s = e - d 2^255, and s = +-d 2^255 (mod n) would need e = d 2^256 (mod n) -- a value far below 2^255, whose top
    // digit is 0 -- or e = 0.  So with e < n (reduced first) the only exceptional cases are the accumulator at infinity (s = 0: every
    // lower digit zero, a lane flag; the sum is then the table point with Z = 1) and a zero digit (the sum is the accumulator).  Returns
    // the homogeneous (X Z : Y : Z^3).  kw: the scalar reduced mod n.
    static MA_DEV void madd(const int32_t* x2, const int32_t* y2, Pt& p) {       // in: X K <= 9, Y K <= 1, Z K <= 3; out: X K = 4, Y K = 1, Z K = 3
        bool hz;
        madd_h<false>(x2, y2, p, hz);
    }
    // HZ: also report H = 0 -- the two points have the same x: P = +-(x2, y2), where this addition fails (or an operand at infinity)
    template <bool HZ>
    static MA_DEV void madd_h(const int32_t* x2, const int32_t* y2, Pt& p, bool& hz) {
        int32_t z1z1[10], u2[10], s2[10], h[10], hh[10], i_[10], j[10], r[10], v[10], t[10];
        F::sqr(p.Z, z1z1);
        F::mul(x2, z1z1, u2);
        F::mul(p.Z, z1z1, t);
        F::mul(y2, t, s2);
        F::sub(u2, p.X, h);         // 10
        if constexpr (HZ) hz = is_zero(h);
        F::sqr(h, hh);
        F::add(hh, hh, i_);
        F::add(i_, i_, i_);         // 4   I = 4 HH
        F::mul(h, i_, j);
        F::sub(s2, p.Y, r);
        F::add(r, r, r);            // 4   r = 2 (S2 - Y1)
        F::mul(p.X, i_, v);
        F::add(p.Z, h, t);          // 13
        F::sqr(t, t);
        F::sub(t, z1z1, t);
        F::sub(t, hh, p.Z);         // 3   Z3 = (Z1 + H)^2 - Z1Z1 - HH
        F::sqr(r, t);
        F::sub(t, j, t);
        F::sub(t, v, t);
        F::sub(t, v, p.X);          // 4   X3 = r^2 - J - 2V
        F::sub(v, p.X, t);          // 5
        F::add(p.Y, p.Y, u2);       // 2
        F::neg(u2, u2);
        F::mul2(r, t, u2, j, p.Y);  //     Y3 = r (V - X3) - 2 Y1 J:  4 x 5 + 2 x 1
    }
    template <class COMB>
    static MA_DEV void mulgen_acc(const uint64_t* kw, Pt& R) {
        F::set_one(R.X);
        F::set_one(R.Y);
        F::zero(R.Z);
        bool rinf = true;
        wn26_mulgen_walk<CvNist256, COMB>(kw, [&](const int32_t* sx, const int32_t* sy, bool zero) {
            Pt S = R;
            madd(sx, sy, S);
            int32_t one[10], u[10];
            F::set_one(one);
            // accumulator at infinity: the sum is (sx, sy, 1); zero digit: the accumulator stays
            F::select(rinf, S.X, sx, u);
            F::select(zero, u, R.X, R.X);
            F::select(rinf, S.Y, sy, u);
            F::select(zero, u, R.Y, R.Y);
            F::select(rinf, S.Z, one, u);
            F::select(zero, u, R.Z, R.Z);
            rinf = rinf && zero;
        });
        to_projective(rinf, R);
    }

    // ---- e P + f Q on two AFFINE tables (wn_affine.h, entries 0..7 and 8..15), the accumulator Jacobian throughout: mixed additions (1 360
    // multiply-adds against the 1 928 of the complete one) and no conversions around the doublings.  The accumulator of a double
    // multiplication can meet +-(table point) anywhere (Q = +-P, f = e ...), which shows as H = 0 in the mixed addition: every addition
    // tests it, and a WAVE in which some lane has it redoes that addition with the complete mixed formula on homogeneous coordinates
    // and takes the result for the lanes concerned.  Its instruction sequence therefore depends on the inputs -- those of a
    // verification, public, as for the reference's own mul2 (weierstrass.c: a joint sparse form with data-dependent branches) and the
    // Straus forms of the Edwards curves (ed26s.h).
    template <class DIG>
    static MA_DEV void mul2_acc_aff(DIG& dige, DIG& digf, const WnAffWs& ws, size_t t, Pt& R) {
        const uint32_t fl = ws.flag[t];
        F::set_one(R.X);
        F::set_one(R.Y);
        F::zero(R.Z);
        bool rinf = true;
#pragma unroll 1
        for (int i = 0; i < 65; i++) {
            if (i != 0) {
#pragma unroll 1
                for (int j = 0; j < 4; j++) dbl(R);
            }
#pragma unroll 1
            for (int which = 0; which < 2; which++) {
                const int dgt = (int)(which ? digf.window(i) : dige.window(i)) - 8;     // [-8, 7]
                const bool neg = dgt < 0;
                const uint32_t m = (uint32_t)(neg ? -dgt : dgt);
                int32_t sx[10], sy[10], one[10], u[10];
                wn_affine_lookup<F>(ws, t, m, neg, sx, sy, 8 * which);
                const bool qinf = ((m == 0) | (((fl >> which) & 1u) != 0)) != 0;
                Pt S = R;
                bool hz;
                madd_h<true>(sx, sy, S, hz);
                bool sinf = false;
                const bool exc = hz && !rinf && !qinf;
#if defined(__HIP_DEVICE_COMPILE__)
                const bool some = __any(exc);
#else
                const bool some = exc;
#endif
                if (some) {                                                             // wave-uniform
                    Pt C = R;
                    to_projective(false, C);
                    E::madd(sx, sy, C);                                                 // complete: the double of the point, or the point at infinity
                    const bool cinf = is_zero(C.Z);
                    from_projective(C);
                    F::select(exc, S.X, C.X, S.X);
                    F::select(exc, S.Y, C.Y, S.Y);
                    F::select(exc, S.Z, C.Z, S.Z);
                    sinf = exc && cinf;
                }
                // R at infinity: the sum is (sx, sy, 1); a zero digit or the table's point at infinity: R stays
                F::set_one(one);
                F::select(rinf, S.X, sx, u);
                F::select(qinf, u, R.X, R.X);
                F::select(rinf, S.Y, sy, u);
                F::select(qinf, u, R.Y, R.Y);
                F::select(rinf, S.Z, one, u);
                F::select(qinf, u, R.Z, R.Z);
                rinf = qinf ? rinf : sinf;
            }
        }
        to_projective(rinf, R);
    }
};

}  // namespace ma
