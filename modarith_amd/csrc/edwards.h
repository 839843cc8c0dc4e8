// modarith_amd/csrc/edwards.h -- Edwards-curve formulas of the batched curve layer for gfx950.
//
// Device-side counterpart of the formulas in the reference's edwards.c: projective points (X:Y:Z) on
// a*x^2 + y^2 = 1 + d*x^2*y^2, a = +-1, from the bit-exact Field<P> functions in the same order as edwards.c, so
// projective limbs equal the reference's (generic=True field) wherever the reference is deterministic.
// Everything curve-independent (cmv, cmp, select, mul, mul2, kernels) is in curve.h.
#pragma once
#include "curve.h"

namespace ma {

template <class C, class F_ = Field<typename C::FieldParams, true>>
struct Edwards : CurveOps<Edwards<C, F_>, typename C::FieldParams, F_> {
    using Base = CurveOps<Edwards<C, F_>, typename C::FieldParams, F_>;
    using P = typename C::FieldParams;
    using F = F_;
    using Point = typename Base::Point;
    using limb_t = typename F::limb_t;
    using Base::cmv;
    using Base::cpy;
    static constexpr int N = P::N, NL = F::NL;
    static constexpr bool HAS_Y_ONLY_SET = true;    // ecnXXXset accepts y + sign of x (edwards.c:362-365)
    static constexpr bool SELECT_FROM_NEUTRAL = true;    // curve.h select(): start the table scan from the neutral element

    // e <- d*e  with the sign handling of edwards.c:81-98
    static MA_DEV void bterm(limb_t* e) {
        if constexpr (C::B_SMALL) {
            F::modmli(e, C::B_INT > 0 ? C::B_INT : -C::B_INT, e);
        } else {
            spint bl[N];
            limb_t b[NL];
            static_for<0, N>([&](auto I) { bl[I] = C::b(I); });
            F::from_limbs(bl, b);
            F::modmul(e, b, e);
        }
    }
    static constexpr bool B_NEG = C::B_SMALL && C::B_INT < 0;

    static MA_DEV void neg(Point& p) { F::modneg(p.x, p.x); }                        // edwards.c:66-69
    static MA_DEV void inf(Point& p) { F::modzer(p.x); F::modone(p.y); F::modone(p.z); }   // edwards.c:171-176
    static MA_DEV int isinf(const Point& p) { return F::modis0(p.x) & F::modcmp(p.y, p.z); }  // edwards.c:179-183
    // P += Q (edwards.c:73-111)
    static MA_DEV void add(const Point& q, Point& p) {
        limb_t A[NL], B[NL], Cc[NL], D[NL], E[NL], Ff[NL], G[NL];
        F::modmul(q.z, p.z, A);
        F::modsqr(A, B);
        F::modmul(q.x, p.x, Cc);
        F::modmul(q.y, p.y, D);
        F::modmul(Cc, D, E);
        bterm(E);
        if constexpr (B_NEG) { F::modadd(B, E, Ff); F::modsub(B, E, G); }
        else                 { F::modsub(B, E, Ff); F::modadd(B, E, G); }
        F::modadd(p.x, p.y, B);
        F::modadd(q.x, q.y, E);
        F::modmul(B, E, p.x);
        F::modsub_u(p.x, Cc, p.x);      // (feeds the next modsub only: field.h "_u")
        F::modsub(p.x, D, p.x);
        F::modmul(p.x, Ff, p.x);
        F::modmul(p.x, A, p.x);
        if constexpr (C::A == -1) F::modadd(D, Cc, p.y); else F::modsub(D, Cc, p.y);
        F::modmul(p.y, A, p.y);
        F::modmul(p.y, G, p.y);
        F::modmul(Ff, G, p.z);
    }
    // P = 2P (edwards.c:123-145)
    static MA_DEV void dbl(Point& p) {
        limb_t B[NL], Cc[NL], D[NL], E[NL], Ff[NL], H[NL], J[NL];
        F::modadd(p.x, p.y, B);
        F::modsqr(B, B);
        F::modsqr(p.x, Cc);
        F::modsqr(p.y, D);
        F::modsqr(p.z, H);
        // the sums that only feed further sums are taken in their "_u" form (field.h: same integer, same sign decision, closing
        // carry chain left to the consumer): 2H -> J, -C -> F and Y3's factor, B - C -> X3's factor
        F::modadd_u(H, H, H);
        if constexpr (C::A == -1) F::modneg_u(Cc, E); else F::modcpy(Cc, E);
        F::modadd(E, D, Ff);
        F::modsub(Ff, H, J);
        F::modsub_u(B, Cc, p.x);
        F::modsub(p.x, D, p.x);
        F::modmul(p.x, J, p.x);
        F::modsub(E, D, p.y);
        F::modmul(p.y, Ff, p.y);
        F::modmul(Ff, J, p.z);
    }
    static MA_DEV void cof(Point& p) {
#pragma unroll 1
        for (int i = 0; i < C::COF; i++) dbl(p);
    }     // edwards.c:338-343

    // edwards.c:186-197; the Z == 0 case is a predicated overwrite instead of an early return
    static MA_DEV void affine(Point& p) {
        spint I[N];
        Point o;
        inf(o);
        const int z0 = F::modis0(p.z);
        F::modinv(p.z, nullptr, I);
        F::modone(p.z);
        F::modmul(p.x, I, p.x);
        F::modmul(p.y, I, p.y);
        cmv(z0, o, p);
    }
    // setxy (edwards.c:246-335).  MODE 0: both coordinates, 1: x and the sign s of y, 2: y and the sign s of x.
    // Off-curve input gives the point at infinity, as there; lane-predicated, no branch on data.
    template <int MODE>
    static MA_DEV void setxy(int s, const spint* x, const spint* y, Point& p) {
        spint X[N], Y[N], O[N], U[N], V[N], H[N];
        Point o;
        inf(o);
        F::modone(O);
        if constexpr (MODE == 0) {
            F::modsqr(x, X);
            F::modsqr(y, Y);
            if constexpr (C::A == -1) F::modsub(Y, X, U); else F::modadd(Y, X, U);
            F::modmul(X, Y, V);
            bterm(V);
            if constexpr (B_NEG) F::modsub(O, V, V); else F::modadd(O, V, V);
            F::modmul(U, O, U);
            F::modmul(V, O, V);
            const int ok = F::modcmp(U, V);
            F::modcpy(x, p.x);
            F::modcpy(y, p.y);
            F::modone(p.z);
            cmv(1 - ok, o, p);
        } else {
            if constexpr (MODE == 1) {
                F::modsqr(x, X);
                if constexpr (C::A == -1) F::modadd(O, X, U); else F::modsub(O, X, U);
                F::modcpy(X, V);
            } else {
                F::modsqr(y, Y);
                F::modsub(O, Y, U);
                if constexpr (C::A == -1) F::modneg(O, O);
                F::modcpy(Y, V);
            }
            bterm(V);
            if constexpr (B_NEG) F::modadd(O, V, V); else F::modsub(O, V, V);
            F::modsqr(U, O);
            F::modmul(U, O, U);
            F::modmul(U, V, U);
            F::modpro(U, H);
            const int ok = F::modqr(H, U);
            F::modsqrt(U, H, V);
            F::modinv(U, H, U);
            F::modmul(U, V, U);
            F::modmul(U, O, U);
            const int d = (F::modsign(U) - s) & 1;
            F::modneg(U, V);
            F::modcmv(d, V, U);
            if constexpr (MODE == 1) { F::modcpy(U, p.y); F::modcpy(x, p.x); }
            else                     { F::modcpy(U, p.x); F::modcpy(y, p.y); }
            F::modone(p.z);
            cmv(1 - ok, o, p);
        }
    }
    static MA_DEV void gen(Point& p) {                                                // edwards.c:369-378
        spint gx[N], gy[N];
        if constexpr (C::SMALL_X != 0) {
            F::modint(C::SMALL_X, gx);
            setxy<1>(0, gx, nullptr, p);
        } else {
            // ecnXXXset(0, x, y) on the generator's own coordinates: the on-curve test of setxy<0> is a fact about the constants
            // (checked where they are emitted and by tests/test_gpu_curveref.py against the reference's ecnXXXgen limbs), so what
            // is left of edwards.c:347-366 is the copy -- a constant store
            static_for<0, N>([&](auto I) { gx[I] = C::gx(I); gy[I] = C::gy(I); });
            F::modcpy(gx, p.x);
            F::modcpy(gy, p.y);
            F::modone(p.z);
        }
    }

};
template <class C, class F_> struct exact_class<Edwards<C, F_>> { using type = Edwards<C, Field<typename C::FieldParams, false>>; };   // curve.h "the limb contract"

}  // namespace ma
