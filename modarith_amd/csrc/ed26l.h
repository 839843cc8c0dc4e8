// modarith_amd/csrc/ed26l.h -- round 5: the fused ED25519 scalar multiplications (ecnXXXmul + ecnXXXget, edwards.c:435-482 +
// 221-239) as a Montgomery LADDER on the birationally equivalent curve, with the v-coordinate recovered at the end.
//
// Only canonical affine bytes leave the fused kernels, so representation and algorithm are free (ed26.h).  The window method
// of ed26.h costs 255 doublings (4S + 3M) + 86 mixed additions (7M) + two inversions per scalar: 2.38e5 multiply-adds.  On
//      M: v^2 = u^3 + A u^2 + u,  A = 486662,          (u, v) = ((1 + y) / (1 - y),  c u / x),   c^2 = -(A + 2)
// (RFC 7748 section 4.1's map; a group isomorphism from the Edwards curve -x^2 + y^2 = 1 + d x^2 y^2 onto M, (0, 1) -> infinity,
// (0, -1) -> (0, 0)) the x-only ladder of rfc7748.c:186-221 takes 5M + 4S + one small multiplication per scalar BIT and needs
// no table, no table scan, no LDS: 256 x 739 = 1.89e5 multiply-adds, the loop of csrc/fe26.h.  The ladder ends with
// (Xq : Zq) = u([e]P) and (Xs : Zs) = u([e + 1]P); with the affine (u, v) of P the v-coordinate of [e]P follows from the
// addition law (Okeya-Sakurai, CHES 2001):
//      v_q = [(u u_q + 1)(u + u_q + 2A) - 2A - (u - u_q)^2 u_s] / (2 v),
// projectively 10M + 1S, and the way back to the Edwards curve is x = c u_q / v_q, y = (u_q - 1) / (u_q + 1).  c itself never
// appears: v enters only as w = 2 c v = 2 c^2 u / x = -2 (A + 2) u / x, a SMALL multiple of coordinates.
//
// Both inversions (the affine (u, w) of P in front, the affine (x, y) of the result at the end) are shared by up to 32 records
// with Montgomery's trick, as the X25519 ladder shares its one (fe_finish.h): four kernels per batch,
//      k_*_lad_prep   P = (X : Y : Z) ->  D = (Z - Y) X,  nu = (Z + Y) X,  nw = -2 (A + 2) (Z + Y) Z          (3M)
//      k_fe_batch_div u = nu / D,  w = nw / D                                                                 (5M + 1/32 inversion)
//      k_*_lad        ladder, recovery, Edwards (X : Y : Z) of [e]P                                           (256 steps + 12M + 1S)
//      k_fe_batch_div x = X / Z,  y = Y / Z, exported as the reference's big-endian records                   (5M + 1/32 inversion)
// = 1.94e5 multiply-adds per scalar, 0.81 of the window method's.
//
// Exceptional cases, all decided by masks on canonical zero tests (no branch, no address depends on them):
//   P in {(0, 1), (0, -1)}  (X = 0: the map has no finite image / v = 0)   ->  [e]P = (e odd and P != neutral) ? (0, -1) : (0, 1)
//   Zq = 0  ([e]P is the neutral element)                                  ->  (0, 1)
//   Xq = 0  ([e]P = (0, -1); the recovered v is 0 and c u / v is 0 / 0)     ->  (0, -1)
//   Zs = 0  ([e + 1]P neutral, i.e. [e]P = -P; the recovery degenerates)   ->  -P = (u, -v) on M, taken back to the Edwards curve
// Every other pair (P, e) has v != 0, v_q != 0 and u_q != -1 (no rational point of M has u = -1: A - 2 is a non-square; the
// other points of order 2 are irrational: A^2 - 4 is a non-square), so no denominator vanishes.  For input points that are not
// on the curve neither this kernel nor the reference's means anything; a record whose denominator comes out zero is exported
// as zeros and cannot disturb the records it shares an inversion with (FeBatchDiv replaces a zero denominator by one).
#pragma once
#include "ed26.h"

namespace ma {

// r = m ? g : f with m a full-lane mask (0 or ~0) the compiler knows nothing about: computed by every lane (field.h lane_mask)
MA_DEV void fe26_blend(uint32_t m, const uint32_t* f, const uint32_t* g, uint32_t* r) {
    static_for<0, 10>([&](auto I) {
        const uint32_t x = f[I], y = g[I];
        r[I] = x ^ ((x ^ y) & m);
    });
}

template <class C>
struct Ed26Lad {
    using F = Fe26;
    using E = Ed26<C>;
    using Ext = typename E::Ext;
    static constexpr uint32_t A2 = 2 * 486662;          // 2A
    static constexpr uint32_t C2 = 2 * 486664;          // -2 c^2 = 2 (A + 2)
    static constexpr uint32_t FLAG_X0 = 1, FLAG_NEUTRAL = 2;

    static MA_DEV bool is_zero(const uint32_t* f) {     // f = 0 mod p, any limbs below 2^32
        uint64_t w[4];
        F::to_words(f, w);
        return (w[0] | w[1] | w[2] | w[3]) == 0;
    }

    // ---- in front of the first shared inversion.  X, Y, Z: the projective point, 5 x 51-bit limbs (field.c form, limbs below 2^53).
    // D, nu, nw: tight.  Returns FLAG_X0 | FLAG_NEUTRAL; with X = 0 the denominator handed on is 1 (nu, nw are then not used).
    static MA_DEV uint32_t prep(const spint* X, const spint* Y, const spint* Z, uint32_t* D, uint32_t* nu, uint32_t* nw) {
        uint32_t px[10], py[10], pz[10], N[10], M[10], t[10], zero[10], one[10];
        E::from51(X, px);
        E::from51(Y, py);
        E::from51(Z, pz);
        F::add(pz, py, N);                  // 1.0
        F::sub(pz, py, M);                  // 1.5
        const bool x0 = is_zero(px), yz = is_zero(M);
        F::mul(M, px, D);
        F::mul(N, px, nu);
        F::mul(N, pz, t);
        F::template mul_small<C2>(t, t);    // tight
        F::set(0, zero);
        F::sub(zero, t, nw);                // 2p - t: 1.0 .. 1.5
        uint32_t wt[10];
        static_for<0, 10>([&](auto I) { wt[I] = nw[I]; });
        E::wc(wt);                          // tight, so that the caller may store limbs or words alike
        static_for<0, 10>([&](auto I) { nw[I] = wt[I]; });
        F::set(1, one);
        fe26_blend((uint32_t)lane_mask(x0), D, one, D);
        return (x0 ? FLAG_X0 : 0u) | (yz ? FLAG_NEUTRAL : 0u);
    }

    // ---- the ladder on M.  ew: the scalar, four little-endian words, any value below 2^256 (ecnXXXmul takes every Nbytes-long
    // scalar, edwards.c:435-482); u: tight limbs of u(P).  Leaves (x2 : z2) = u([e]P) and (x3 : z3) = u([e + 1]P), tight.
    static MA_DEV void ladder(const uint64_t* ew, const uint32_t* u, uint32_t* x2, uint32_t* z2, uint32_t* x3, uint32_t* z3) {
        uint64_t kw[4];
        static_for<0, 4>([&](auto K) { kw[K] = ew[K]; });
        uint32_t x1_19[10];
        F::pre19(u, x1_19);
        x1_19[0] = 0;
        F::set(1, x2);
        F::set(0, z2);
        F::copy(u, x3);
        F::set(1, z3);
        uint32_t swap = 0;
        // the step of fe26.h x25519_fe26_ladder (selects instead of swaps: the pair {DA, CB} does not see the swap), 256 bits, no clamp
#pragma unroll 1
        for (int step = 0; step < 256; step++) {
            const uint32_t kt = (uint32_t)(kw[3] >> 63);
            kw[3] = (kw[3] << 1) | (kw[2] >> 63);
            kw[2] = (kw[2] << 1) | (kw[1] >> 63);
            kw[1] = (kw[1] << 1) | (kw[0] >> 63);
            kw[0] <<= 1;
            const bool sw = (swap ^ kt) != 0;
            swap = kt;
            uint32_t A[10], B[10], Cc[10], D[10], As[10], Bs[10], AA[10], BB[10], Ee[10];
            F::add(x2, z2, A);
            F::add(x3, z3, Cc);
            F::sub(x2, z2, B);
            F::sub(x3, z3, D);
            F::select(sw, A, Cc, As);
            F::select(sw, B, D, Bs);
            F::mul(D, A, D);
            F::mul(Cc, B, Cc);
            F::sqr(As, AA);
            F::sqr(Bs, BB);
            F::sub(D, Cc, z3);
            F::add(D, Cc, x3);
            F::sub(AA, BB, Ee);
            F::template mul_small_add<121665>(Ee, AA, z2);
            F::mul(z2, Ee, z2);
            F::sqr(x3, x3);
            F::sqr(z3, z3);
            F::mul(z3, u, x1_19, z3);
            F::mul(AA, BB, x2);
        }
        // "2" = [e]P, "3" = [e + 1]P
        const uint32_t m = (uint32_t)lane_mask(swap != 0);
        F::cswap(m, x2, x3);
        F::cswap(m, z2, z3);
    }

    // ---- the way back: v([e]P) recovered from (u, w) of P and the two ladder outputs, then the Edwards point.  e_odd: bit 0 of the
    // scalar; flags: prep()'s.  R = [e]P in extended Edwards coordinates (X : Y : Z : T), T = X Y / Z only when WANT_T; limbs tight.
    template <bool WANT_T>
    static MA_DEV void recover(const uint32_t* u, const uint32_t* w, uint32_t flags, bool e_odd,
                               const uint32_t* x2, const uint32_t* z2, const uint32_t* x3, const uint32_t* z3, Ext& R) {
        // (all operands within one add/sub of tight, as mul / sqr accept)
        uint32_t t1[10], t2[10], a1[10], a2[10], zA[10], m1[10], m2[10], d[10], V[10], K[10], a[10], s[10], m[10];
        F::mul(u, z2, t1);                  // u Zq
        F::mul(u, x2, t2);                  // u Xq
        F::add(t2, z2, a1);                 // 1.0   u Xq + Zq
        F::template mul_small<A2>(z2, zA);  // tight 2A Zq
        F::add(t1, x2, a2);
        F::add(a2, zA, a2);                 // 1.5   u Zq + Xq + 2A Zq
        F::mul(a1, a2, m1);
        F::mul(zA, z2, m2);                 // 2A Zq^2
        F::sub(m1, m2, m1);                 // 1.5
        F::mul(m1, z3, m1);
        F::sub(t1, x2, d);                  // 1.5   u Zq - Xq
        F::sqr(d, d);
        F::mul(d, x3, d);
        F::sub(m1, d, V);                   // 1.5   numerator of v_q over 2 v Zq^2 Zs
        F::mul(w, z2, K);
        F::mul(K, z3, K);                   // c (2 v Zq Zs)
        F::mul(K, x2, a);                   // x = a / V
        F::add(x2, z2, s);                  // 1.0
        F::sub(x2, z2, m);                  // 1.5   y = m / s
        // ---- exceptional cases
        const bool zq0 = is_zero(z2), xq0 = is_zero(x2), zs0 = is_zero(z3);
        const bool px0 = (flags & FLAG_X0) != 0, pn = (flags & FLAG_NEUTRAL) != 0;
        const bool r_neutral = px0 ? (pn || !e_odd) : zq0;
        const bool r_two = px0 ? (!pn && e_odd) : (xq0 && !zq0);
        const bool r_negp = !px0 && zs0 && !zq0 && !xq0;
        uint32_t one[10], zero[10], mone[10], na[10], ns[10], nm[10];
        F::set(1, one);
        F::set(0, zero);
        F::sub(zero, one, mone);            // 2p - 1
        F::template mul_small<C2>(u, na);   // -P = (u, -v):  x = -c u / v = 2 (A + 2) u / w,  y = (u - 1) / (u + 1)
        F::add(u, one, ns);
        F::sub(u, one, nm);                 // 1.5
        const uint32_t mk_n = (uint32_t)lane_mask(r_negp), mk_01 = (uint32_t)lane_mask(r_neutral || r_two), mk_2 = (uint32_t)lane_mask(r_two);
        fe26_blend(mk_n, a, na, a);
        fe26_blend(mk_n, s, ns, s);
        fe26_blend(mk_n, m, nm, m);
        fe26_blend(mk_n, V, w, V);
        fe26_blend(mk_01, a, zero, a);      // (0, +-1): a = 0, V = s = 1, m = +-1
        fe26_blend(mk_01, s, one, s);
        fe26_blend(mk_01, m, one, m);
        fe26_blend(mk_01, V, one, V);
        fe26_blend(mk_2, m, mone, m);
        F::mul(a, s, R.X);
        F::mul(V, m, R.Y);
        F::mul(V, s, R.Z);
        if constexpr (WANT_T) F::mul(a, m, R.T);
    }
    template <bool WANT_T>
    static MA_DEV void mul(const uint64_t* ew, const uint32_t* u, const uint32_t* w, uint32_t flags, Ext& R) {
        uint32_t x2[10], z2[10], x3[10], z3[10];
        ladder(ew, u, x2, z2, x3, z3);
        recover<WANT_T>(u, w, flags, (ew[0] & 1) != 0, x2, z2, x3, z3, R);
    }

    // One fused scalar multiplication + affine export with its own two inversions: the per-lane reference of the kernel pipeline
    // (tools/fe_host_check.hip runs it on the host against the oracle's ecn mul + get).
    static MA_DEV void mul_get_one(const uint64_t* ew, const spint* X, const spint* Y, const spint* Z, uint64_t* xw, uint64_t* yw) {
        uint32_t D[10], nu[10], nw[10], u[10], w[10];
        const uint32_t flags = prep(X, Y, Z, D, nu, nw);
        F::invert(D, D);
        F::mul(nu, D, u);
        F::mul(nw, D, w);
        Ext R;
        mul<false>(ew, u, w, flags, R);
        uint32_t zi[10], ax[10], ay[10];
        F::invert(R.Z, zi);
        F::mul(R.X, zi, ax);
        F::mul(R.Y, zi, ay);
        F::to_words(ax, xw);
        F::to_words(ay, yw);
    }
    // e*G + f*Q (the verification pattern, ed26.h ed25519_mulgen2_get_one): f*Q by the ladder, e*G added through the fixed-base
    // table with complete mixed additions
    template <class TAB>
    static MA_DEV void mulgen2_acc(const uint64_t* ew, const uint64_t* fw, const uint32_t* u, const uint32_t* w, uint32_t flags, Ext& R) {
        mul<true>(fw, u, w, flags, R);
        ed25519_mulgen_acc<C, TAB, false>(ew, R);
    }
    template <class TAB>
    static MA_DEV void mulgen2_get_one(const uint64_t* ew, const uint64_t* fw, const spint* X, const spint* Y, const spint* Z, uint64_t* xw, uint64_t* yw) {
        uint32_t D[10], nu[10], nw[10], u[10], w[10];
        const uint32_t flags = prep(X, Y, Z, D, nu, nw);
        F::invert(D, D);
        F::mul(nu, D, u);
        F::mul(nw, D, w);
        Ext R;
        mulgen2_acc<TAB>(ew, fw, u, w, flags, R);
        uint32_t zi[10], ax[10], ay[10];
        F::invert(R.Z, zi);
        F::mul(R.X, zi, ax);
        F::mul(R.Y, zi, ay);
        F::to_words(ax, xw);
        F::to_words(ay, yw);
    }
};

}  // namespace ma
