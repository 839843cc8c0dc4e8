// modarith_amd/csrc/weierstrass.h -- short-Weierstrass formulas of the batched curve layer.
//
// Device-side counterpart of the formulas in the reference's weierstrass.c: projective points (X:Y:Z) on
// y^2 = x^3 + a*x + b with the complete addition / doubling of Renes-Costello-Batina (eprint 2015/1060) in the
// operation order of weierstrass.c:68-281 (a = -3 and a = 0 branches; b as a limb constant or, below 2^28 in magnitude,
// as the C int CONSTANT_B multiplied in with modmli; small CONSTANT_X generators), built from the
// bit-exact Field<P> functions.  Everything curve-independent is in curve.h.
#pragma once
#include "curve.h"

namespace ma {

template <class C, class F_ = Field<typename C::FieldParams, true>>     // F_: the limb form, or a resident half-limb form for the scalar multiplications (tried for P-256 in round 4 and dropped: docs/curve_layer.md)
struct Weierstrass : CurveOps<Weierstrass<C, F_>, typename C::FieldParams, F_> {
    using Base = CurveOps<Weierstrass<C, F_>, typename C::FieldParams, F_>;
    using P = typename C::FieldParams;
    using F = F_;
    using Point = typename Base::Point;
    using limb_t = typename F::limb_t;
    using Base::cmv;
    using Base::cpy;
    static constexpr int N = P::N, NL = F::NL;
    static constexpr bool HAS_Y_ONLY_SET = false;   // weierstrass.c:417-428: x is mandatory
    static constexpr bool SELECT_FROM_NEUTRAL = (P::N < 9);    // curve.h select(): start the table scan from the neutral element

    static MA_DEV void const_b(limb_t* b) { spint l[N]; static_for<0, N>([&](auto I) { l[I] = C::b(I); }); F::from_limbs(l, b); }
    static MA_DEV void const_b3(limb_t* b) { spint l[N]; static_for<0, N>([&](auto I) { l[I] = C::b3(I); }); F::from_limbs(l, b); }

    static MA_DEV void neg(Point& p) { F::modneg(p.y, p.y); }                          // weierstrass.c:61-64
    static MA_DEV void inf(Point& p) { F::modzer(p.x); F::modone(p.y); F::modzer(p.z); }  // weierstrass.c:284-289
    static MA_DEV int isinf(const Point& p) { return F::modis0(p.x) & F::modis0(p.z); }   // weierstrass.c:292-296
    static MA_DEV void cof(Point&) {}                                                  // weierstrass.c:413-414

    // P += Q, complete (weierstrass.c:68-175)
    static MA_DEV void add(const Point& q, Point& p) {
        limb_t B[NL], T0[NL], T1[NL], T2[NL], T3[NL], T4[NL];
        F::modmul(p.x, q.x, T0);
        F::modmul(p.y, q.y, T1);
        F::modmul(p.z, q.z, T2);
        F::modadd(p.x, p.y, T3);
        F::modadd(q.x, q.y, T4);
        F::modmul(T3, T4, T3);
        F::modadd_u(T0, T1, T4);          // "_u" (field.h): sums that only feed further sums leave their closing carry chain to the consumer
        F::modsub(T3, T4, T3);
        F::modadd(p.y, p.z, T4);
        F::modadd(q.y, q.z, B);
        F::modmul(T4, B, T4);
        F::modadd_u(T1, T2, B);
        F::modsub(T4, B, T4);
        F::modadd(p.x, p.z, p.x);
        F::modadd(q.z, q.x, p.y);
        F::modmul(p.x, p.y, p.x);
        F::modadd_u(T0, T2, p.y);
        F::modsub(p.x, p.y, p.y);
        if constexpr (C::A == 0) {
            F::modadd(T0, T0, p.x);
            F::modadd(T0, p.x, T0);
            if constexpr (C::SMALL_B > 0) {
                F::modmli(T2, 3 * C::SMALL_B, T2);
                F::modmli(p.y, 3 * C::SMALL_B, p.y);
            } else if constexpr (C::SMALL_B < 0) {
                F::modmli(T2, -3 * C::SMALL_B, T2); F::modneg(T2, T2);
                F::modmli(p.y, -3 * C::SMALL_B, p.y); F::modneg(p.y, p.y);
            } else {
                const_b3(B);
                F::modmul(T2, B, T2);
                F::modmul(p.y, B, p.y);
            }
            F::modadd(T1, T2, p.z);
            F::modsub(T1, T2, T1);
            F::modmul(p.y, T4, p.x);
            F::modmul(T3, T1, T2);
            F::modsub(T2, p.x, p.x);
            F::modmul(p.y, T0, p.y);
            F::modmul(T1, p.z, T1);
            F::modadd(p.y, T1, p.y);
            F::modmul(T0, T3, T0);
            F::modmul(p.z, T4, p.z);
            F::modadd(p.z, T0, p.z);
        } else {
            static_assert(C::A == 0 || C::A == -3, "weierstrass.c handles a = 0 and a = -3");
            if constexpr (C::SMALL_B > 0) {
                F::modmli(T2, C::SMALL_B, p.z);
                F::modsub(p.y, p.z, p.x);
                F::modmli(p.y, C::SMALL_B, p.y);
            } else if constexpr (C::SMALL_B < 0) {
                F::modmli(T2, -C::SMALL_B, p.z);
                F::modadd(p.y, p.z, p.x);
                F::modmli(p.y, -C::SMALL_B, p.y); F::modneg(p.y, p.y);
            } else {
                const_b(B);
                F::modmul(B, T2, p.z);
                F::modsub_u(p.y, p.z, p.x);     // -> 2x, 3x below
                F::modmul(p.y, B, p.y);
            }
            F::modadd_u(p.x, p.x, p.z);         // 2x -> 3x
            F::modadd_u(p.x, p.z, p.x);         // 3x -> T1 - 3x, 3x + T1
            F::modsub(T1, p.x, p.z);
            F::modadd(p.x, T1, p.x);
            F::modadd_u(T2, T2, T1);            // 2 T2 -> 3 T2
            F::modadd_u(T2, T1, T2);            // 3 T2 -> two differences
            F::modsub_u(p.y, T2, p.y);          // -> next difference
            F::modsub_u(p.y, T0, p.y);          // -> 2y, 3y
            F::modadd_u(p.y, p.y, T1);
            F::modadd(p.y, T1, p.y);
            F::modadd_u(T0, T0, T1);
            F::modadd_u(T0, T1, T0);
            F::modsub(T0, T2, T0);
            F::modmul(T4, p.y, T1);
            F::modmul(T0, p.y, T2);
            F::modmul(p.x, p.z, p.y);
            F::modadd(p.y, T2, p.y);
            F::modmul(p.x, T3, p.x);
            F::modsub(p.x, T1, p.x);
            F::modmul(p.z, T4, p.z);
            F::modmul(T3, T0, T1);
            F::modadd(p.z, T1, p.z);
        }
    }

    // P = 2P, complete (weierstrass.c:187-281)
    static MA_DEV void dbl(Point& p) {
        limb_t B[NL], T0[NL], T1[NL], T2[NL], T3[NL], T4[NL];
        if constexpr (C::A == 0) {
            F::modsqr(p.y, T0);
            F::modadd(T0, T0, T3);
            F::modadd(T3, T3, T3);
            F::modadd(T3, T3, T3);
            F::modmul(p.x, p.y, T4);
            F::modmul(p.y, p.z, T1);
            F::modsqr(p.z, T2);
            if constexpr (C::SMALL_B > 0) {
                F::modmli(T2, 3 * C::SMALL_B, T2);
            } else if constexpr (C::SMALL_B < 0) {
                F::modmli(T2, -3 * C::SMALL_B, T2); F::modneg(T2, T2);
            } else {
                const_b3(B);
                F::modmul(T2, B, T2);
            }
            F::modmul(T2, T3, p.x);
            F::modadd(T0, T2, p.y);
            F::modmul(T3, T1, p.z);
            F::modadd(T2, T2, T1);
            F::modadd(T2, T1, T2);
            F::modsub(T0, T2, T0);
            F::modmul(p.y, T0, p.y);
            F::modadd(p.y, p.x, p.y);
            F::modmul(T0, T4, p.x);
            F::modadd(p.x, p.x, p.x);
        } else {
            F::modsqr(p.x, T0);
            F::modsqr(p.y, T1);
            F::modsqr(p.z, T2);
            F::modmul(p.x, p.y, T3);
            F::modmul(p.y, p.z, T4);
            F::modadd(T3, T3, T3);
            F::modmul(p.z, p.x, p.z);
            F::modadd(p.z, p.z, p.z);
            if constexpr (C::SMALL_B > 0) {
                F::modmli(T2, C::SMALL_B, p.y);
                F::modsub(p.y, p.z, p.y);
                F::modmli(p.z, C::SMALL_B, p.z);
            } else if constexpr (C::SMALL_B < 0) {
                F::modmli(T2, -C::SMALL_B, p.y); F::modneg(p.y, p.y);
                F::modsub(p.y, p.z, p.y);
                F::modmli(p.z, -C::SMALL_B, p.z); F::modneg(p.z, p.z);
            } else {
                const_b(B);
                F::modmul(T2, B, p.y);
                F::modsub_u(p.y, p.z, p.y);     // "_u" (field.h): -> 2y, 3y below
                F::modmul(p.z, B, p.z);
            }
            F::modadd_u(p.y, p.y, p.x);
            F::modadd_u(p.y, p.x, p.y);
            F::modsub(T1, p.y, p.x);
            F::modadd(p.y, T1, p.y);
            F::modmul(p.y, p.x, p.y);
            F::modmul(p.x, T3, p.x);
            F::modadd_u(T2, T2, T3);
            F::modadd_u(T2, T3, T2);
            F::modsub_u(p.z, T2, p.z);
            F::modsub_u(p.z, T0, p.z);
            F::modadd_u(p.z, p.z, T3);
            F::modadd(p.z, T3, p.z);
            F::modadd_u(T0, T0, T3);
            F::modadd_u(T0, T3, T0);
            F::modsub(T0, T2, T0);
            F::modmul(T0, p.z, T0);
            F::modadd(p.y, T0, p.y);
            F::modadd(T4, T4, T4);
            F::modmul(p.z, T4, p.z);
            F::modsub(p.x, p.z, p.x);
            F::modmul(T4, T1, p.z);
            F::modadd_u(p.z, p.z, p.z);
            F::modadd(p.z, p.z, p.z);
        }
    }

    // weierstrass.c:299-310; Z == 0 handled by a predicated overwrite with (0 : 1 : 0)
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

    // setxy (weierstrass.c:366-410).  MODE 0: (x, y); MODE 1: x and the sign s of y.  Off-curve -> infinity.
    template <int MODE>
    static MA_DEV void setxy(int s, const spint* x, const spint* y, Point& p) {
        static_assert(MODE == 0 || MODE == 1, "weierstrass.c sets a point from x (and optionally y)");
        spint T[N], V[N], H[N], B[N];
        Point o;
        inf(o);
        F::modcpy(x, p.x);
        F::modsqr(x, V);
        F::modmul(V, x, V);
        if constexpr (C::A == -3) {
            F::modsub(V, x, V);
            F::modsub(V, x, V);
            F::modsub(V, x, V);
        }
        if constexpr (C::SMALL_B > 0) {
            F::modint(C::SMALL_B, B);
            F::modadd(V, B, V);
        } else if constexpr (C::SMALL_B < 0) {
            F::modint(-C::SMALL_B, B);
            F::modsub(V, B, V);
        } else {
            const_b(B);
            F::modadd(V, B, V);
        }
        if constexpr (MODE == 0) {
            F::modsqr(y, T);
            const int ok = F::modcmp(T, V);
            F::modcpy(y, p.y);
            F::modone(p.z);
            cmv(1 - ok, o, p);
        } else {
            F::modpro(V, H);
            const int ok = F::modqr(H, V);
            F::modsqrt(V, H, p.y);
            const int d = (F::modsign(p.y) - s) & 1;
            F::modneg(p.y, T);
            F::modcmv(d, T, p.y);
            F::modone(p.z);
            cmv(1 - ok, o, p);
        }
    }
    static MA_DEV void gen(Point& p) {                                                  // weierstrass.c:431-440
        spint gx[N], gy[N];
        if constexpr (C::SMALL_X != 0) {
            F::modint(C::SMALL_X, gx);
            setxy<1>(0, gx, nullptr, p);
        } else {
            // ecnXXXset(0, x, y) on the generator's own coordinates: the on-curve test of setxy<0> is a fact about the constants
            // (checked where they are emitted and by tests/test_gpu_curveref.py against the reference's ecnXXXgen limbs), so what
            // is left of weierstrass.c:417-428 is the copy -- a constant store
            static_for<0, N>([&](auto I) { gx[I] = C::gx(I); gy[I] = C::gy(I); });
            F::modcpy(gx, p.x);
            F::modcpy(gy, p.y);
            F::modone(p.z);
        }
    }
};
template <class C, class F_> struct exact_class<Weierstrass<C, F_>> { using type = Weierstrass<C, Field<typename C::FieldParams, false>>; };   // curve.h "the limb contract"

}  // namespace ma
