// modarith_amd/csrc/ed28l.h -- round 5: the fused ED448 scalar multiplications (ecnXXXmul + ecnXXXget, edwards.c:435-482 + 221-239) as
// a Montgomery LADDER on the birationally equivalent curve with the second coordinate recovered at the end: the construction of
// csrc/ed26l.h (which see for the derivation and the exceptional cases) on the fe28 representation.
//
// ED448 is x^2 + y^2 = 1 + d x^2 y^2 with a = 1, d = -39081 (curve.py:97-105).  The birationally equivalent Montgomery curve is
//      M: B v^2 = u^3 + A u^2 + u,    A = 2 (a + d) / (a - d) = -39080 / 19541,    B = 4 / (a - d) = 2 / 19541,
//      (u, v) = ((1 + y) / (1 - y), u / x),          (x, y) = (u / v, (u - 1) / (u + 1))
// (NOT curve448 of RFC 7748, which is 4-isogenous to this curve, not isomorphic).  Its ladder constant (A + 2) / 4 = a / (a - d)
// = 1 / 39082 is not small, but the doubling only needs the RATIO x2' : z2' = AA BB : E (BB + E / 39082) = AA (39082 BB) :
// E (39082 BB + E): one small multiplication per step, exactly the cost of the X448 step (fe28.h x448_fe28_ladder).  448 steps for
// the 448-bit scalars of ecnXXXmul.  Recovery with A = An / Ad, An = -39080, Ad = 19541, B Ad = 2:
//      v_q = [ (u u_q + 1)(Ad (u + u_q) + 2 An) - 2 An - Ad (u - u_q)^2 u_s ] / (4 v)
// and x = u_q / v_q, y = (u_q - 1) / (u_q + 1); v enters as w = 4 v = 4 u / x = 4 (Z + Y) Z / ((Z - Y) X).
// Exceptional cases as in ed26l.h (the group is Z/4 x Z/q: neutral (0, 1), order two (0, -1), order four (+-1, 0); M has no rational
// point with u = -1 and no other rational point of order two because d is a non-square).
#pragma once
#include "ed28.h"

namespace ma {

MA_DEV void fe28_blend(uint32_t m, const uint32_t* f, const uint32_t* g, uint32_t* r) {
    static_for<0, 16>([&](auto I) {
        const uint32_t x = f[I], y = g[I];
        r[I] = x ^ ((x ^ y) & m);
    });
}

struct Ed28Lad {
    using F = Fe28;
    using E = Ed28;
    using Ext = E::Ext;
    static constexpr uint32_t AD = 19541, AN2 = 2 * 39080;      // A = -39080 / 19541
    static constexpr uint32_t FLAG_X0 = 1, FLAG_NEUTRAL = 2;

    static MA_DEV bool is_zero(const uint32_t* f) {     // f = 0 mod p, limbs below 2^30
        uint64_t w[7];
        F::to_words(f, w);
        uint64_t any = 0;
        static_for<0, 7>([&](auto K) { any |= w[K]; });
        return any == 0;
    }

    // in front of the first shared inversion: D = (Z - Y) X, nu = (Z + Y) X, nw = 4 (Z + Y) Z; all tight.  X = 0 hands on D = 1.
    static MA_DEV uint32_t prep(const spint* X, const spint* Y, const spint* Z, uint32_t* D, uint32_t* nu, uint32_t* nw) {
        uint32_t px[16], py[16], pz[16], N[16], M[16], t[16], one[16];
        E::from56(X, px);
        E::from56(Y, py);
        E::from56(Z, pz);
        F::add(pz, py, N);                  // < 2^29
        F::sub(pz, py, M);                  // tight
        const bool x0 = is_zero(px), yz = is_zero(M);
        F::mul_k(M, px, D);
        F::mul_k(px, N, nu);                // (one operand tight, the other below 2^29)
        F::mul_k(pz, N, t);
        F::template mul_small<4>(t, nw);    // tight
        F::set(1, one);
        fe28_blend((uint32_t)lane_mask(x0), D, one, D);
        return (x0 ? FLAG_X0 : 0u) | (yz ? FLAG_NEUTRAL : 0u);
    }

    // the ladder on M: ew = the scalar, seven little-endian words, any value below 2^448; u tight.  Leaves (x2 : z2) = u([e]P),
    // (x3 : z3) = u([e + 1]P), tight.
    static MA_DEV void ladder(const uint64_t* ew, const uint32_t* u, uint32_t* x2, uint32_t* z2, uint32_t* x3, uint32_t* z3) {
        uint64_t kw[7];
        static_for<0, 7>([&](auto K) { kw[K] = ew[K]; });
        F::set(1, x2);
        F::set(0, z2);
        F::copy(u, x3);
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
            uint32_t A[16], B[16], C[16], D[16], As[16], Bs[16], AA[16], BB[16], Ee[16], t[16];
            F::add(x2, z2, A);
            F::add(x3, z3, C);
            F::sub(x2, z2, B);
            F::sub(x3, z3, D);
            F::select(sw, A, C, As);
            F::select(sw, B, D, Bs);
            F::mul_k(D, A, D);                  // D, B tight; A, C below 2^29
            F::mul_k(C, B, C);
            F::sqr(As, AA);
            F::sqr_k(Bs, BB);
            F::sub(D, C, z3);
            F::add(D, C, x3);
            F::sub(AA, BB, Ee);                 // tight
            F::template mul_small<39082>(BB, t);    // tight: (a - d) BB
            F::add(t, Ee, z2);                  // < 2^29
            F::mul_k(Ee, z2, z2);               // E ((a - d) BB + E)
            F::sqr(x3, x3);
            F::sqr_k(z3, z3);
            F::mul_k(z3, u, z3);
            F::mul_k(t, AA, x2);                // (a - d) AA BB
        }
        const uint32_t m = (uint32_t)lane_mask(swap != 0);
        F::cswap(m, x2, x3);
        F::cswap(m, z2, z3);
    }

    // the way back; R = [e]P in extended Edwards coordinates, T only when want_t (a compile-time or wave-uniform flag).  u and w are
    // FETCHED where they are used (load_u(out), load_w(out): the kernels re-read them from the workspace rows) and the ladder's
    // outputs are consumed in an order that lets each die early -- held across the whole recovery, the six 16-limb inputs plus a
    // product's own columns do not fit the 256 registers of the ladder kernel (34 spilled).
    template <class LU, class LW>
    static MA_DEV void recover(LU load_u, LW load_w, uint32_t flags, bool e_odd,
                               const uint32_t* x2, const uint32_t* z2, const uint32_t* x3, const uint32_t* z3, Ext& R, bool want_t) {
        uint32_t K[16], V[16], a[16], s[16], m[16];
        const bool zq0 = is_zero(z2), xq0 = is_zero(x2), zs0 = is_zero(z3);
        {
            uint32_t w[16];
            load_w(w);
            F::mul_k(w, z2, K);
            F::mul_k(K, z3, K);                 // 4 v Zq Zs
        }
        {
            uint32_t u[16], t1[16], d[16], m1[16];
            load_u(u);
            F::mul_k(u, z2, t1);                // u Zq
            F::sub(t1, x2, d);                  // tight       u Zq - Xq
            F::sqr_k(d, d);
            F::template mul_small<AD>(d, d);
            F::mul_k(d, x3, d);                 // Ad (u Zq - Xq)^2 Xs                                   (x3 dies)
            {
                uint32_t t2[16], a1[16], a2[16], zA[16], m2[16];
                F::mul_k(u, x2, t2);            // u Xq                                                  (u dies)
                F::add(t2, z2, a1);             // < 2^29      u Xq + Zq
                F::add(t1, x2, a2);             // < 2^29
                F::template mul_small<AD>(a2, a2);      // tight       Ad (u Zq + Xq)
                F::template mul_small<AN2>(z2, zA);     // tight       |2 An| Zq
                F::sub(a2, zA, a2);             // tight       Ad (u Zq + Xq) + 2 An Zq
                F::mul_k(a2, a1, m1);
                F::mul_k(zA, z2, m2);           // |2 An| Zq^2
                F::add(m1, m2, m1);             // < 2^29      ... - 2 An Zq^2
            }
            F::mul_k(z3, m1, m1);               //                                                       (z3 dies)
            F::sub(m1, d, V);                   // tight       numerator of v_q over 4 v Zq^2 Zs
        }
        F::mul_k(K, x2, a);                     // x = a / V
        F::add(x2, z2, s);                      // < 2^29
        F::sub(x2, z2, m);                      // tight       y = m / s
        const bool px0 = (flags & FLAG_X0) != 0, pn = (flags & FLAG_NEUTRAL) != 0;
        const bool r_neutral = px0 ? (pn || !e_odd) : zq0;
        const bool r_two = px0 ? (!pn && e_odd) : (xq0 && !zq0);
        const bool r_negp = !px0 && zs0 && !zq0 && !xq0;
        const uint32_t mk_n = (uint32_t)lane_mask(r_negp), mk_01 = (uint32_t)lane_mask(r_neutral || r_two), mk_2 = (uint32_t)lane_mask(r_two);
        uint32_t one[16], zero[16];
        F::set(1, one);
        F::set(0, zero);
        {   // -P = (u, -v):  x = u / (-v) = -4 u / w,  y = (u - 1) / (u + 1)
            uint32_t u[16], t[16];
            load_u(u);
            F::template mul_small<4>(u, t);
            F::sub(zero, t, t);
            fe28_blend(mk_n, a, t, a);
            F::add(u, one, t);
            fe28_blend(mk_n, s, t, s);
            F::sub(u, one, t);
            fe28_blend(mk_n, m, t, m);
            load_w(t);
            fe28_blend(mk_n, V, t, V);
        }
        fe28_blend(mk_01, a, zero, a);          // (0, +-1): a = 0, V = s = 1, m = +-1
        fe28_blend(mk_01, s, one, s);
        fe28_blend(mk_01, m, one, m);
        fe28_blend(mk_01, V, one, V);
        {
            uint32_t mone[16];
            F::sub(zero, one, mone);            // p - 1, tight
            fe28_blend(mk_2, m, mone, m);
        }
        // a, V, m tight; s below 2^29.  Each product overwrites an operand that dies with it (mul_k writes its result last): four
        // results next to four operands and a product's own columns would not fit
        if (want_t) F::mul_k(a, m, R.T);
        F::mul_k(a, s, a);
        F::mul_k(V, m, m);
        F::mul_k(V, s, V);
        F::copy(a, R.X);
        F::copy(m, R.Y);
        F::copy(V, R.Z);
    }

    // e*G + f*Q with its own two inversions (host check): f*Q by the ladder, e*G added through the fixed-base table (ed28.h ed448_mulgen_acc)
    template <class TAB>
    static MA_DEV void mulgen2_get_one(const uint64_t* ew, const uint64_t* fw, const spint* X, const spint* Y, const spint* Z, uint64_t* xw, uint64_t* yw) {
        uint32_t D[16], nu[16], nw[16], u[16], w[16], x2[16], z2[16], x3[16], z3[16];
        const uint32_t flags = prep(X, Y, Z, D, nu, nw);
        F::invert(D, D);
        F::mul_k(nu, D, u);
        F::mul_k(nw, D, w);
        ladder(fw, u, x2, z2, x3, z3);
        Ext R;
        recover([&](uint32_t* o) { F::copy(u, o); }, [&](uint32_t* o) { F::copy(w, o); }, flags, (fw[0] & 1) != 0, x2, z2, x3, z3, R, true);
        ed448_mulgen_acc<TAB, false>(ew, R);
        uint32_t zi[16], ax[16], ay[16];
        F::invert(R.Z, zi);
        F::mul_k(R.X, zi, ax);
        F::mul_k(R.Y, zi, ay);
        F::to_words(ax, xw);
        F::to_words(ay, yw);
    }
    // one fused multiplication + export with its own two inversions: the per-lane reference of the kernel pipeline (host check)
    static MA_DEV void mul_get_one(const uint64_t* ew, const spint* X, const spint* Y, const spint* Z, uint64_t* xw, uint64_t* yw) {
        uint32_t D[16], nu[16], nw[16], u[16], w[16], x2[16], z2[16], x3[16], z3[16];
        const uint32_t flags = prep(X, Y, Z, D, nu, nw);
        F::invert(D, D);
        F::mul_k(nu, D, u);
        F::mul_k(nw, D, w);
        ladder(ew, u, x2, z2, x3, z3);
        Ext R;
        recover([&](uint32_t* o) { F::copy(u, o); }, [&](uint32_t* o) { F::copy(w, o); }, flags, (ew[0] & 1) != 0, x2, z2, x3, z3, R, false);
        uint32_t zi[16], ax[16], ay[16];
        F::invert(R.Z, zi);
        F::mul_k(R.X, zi, ax);
        F::mul_k(R.Y, zi, ay);
        F::to_words(ax, xw);
        F::to_words(ay, yw);
    }
};

}  // namespace ma
