// modarith_amd/csrc/edwards.h -- batched Edwards-curve layer on the field path for gfx950.
//
// Device-side counterpart of the reference's edwards.c (API curve.h:13-29): projective points (X:Y:Z)
// on a*x^2 + y^2 = 1 + d*x^2*y^2, a = +-1, one point per lane with all coordinates in VGPRs, built from
// the bit-exact Field<P> functions in the same order as edwards.c, so projective limbs equal the
// reference's (generic=True field) wherever the reference is deterministic.  The 9-entry window table of
// ecnXXXmul does not fit the register file (9 x 3 x N limbs); it lives in a per-lane slot of a global
// workspace laid out [entry][coord][limb][lane] (coalesced; sized to the resident grid so it stays in the
// 256 MiB Infinity Cache) and is scanned in full on every lookup: the table index never forms an
// address, and selection is lane-predicated modcmv (constant time, edwards.c:381-401).
#pragma once
#include "field.h"
#include "kernels.h"

namespace ma {

template <class C>
struct Edwards {
    using P = typename C::FieldParams;
    using F = Field<P>;
    static constexpr int N = P::N;
    static constexpr int NB = P::NBYTES;
    static constexpr int NW = NB / 8;
    struct Point { spint x[N], y[N], z[N]; };

    // e <- d*e  with the sign handling of edwards.c:81-98
    static MA_DEV void bterm(spint* e) {
        if constexpr (C::B_SMALL) {
            F::modmli(e, C::B_INT > 0 ? C::B_INT : -C::B_INT, e);
        } else {
            spint b[N];
            static_for<0, N>([&](auto I) { b[I] = C::b(I); });
            F::modmul(e, b, e);
        }
    }
    static constexpr bool B_NEG = C::B_SMALL && C::B_INT < 0;

    static MA_DEV void cpy(const Point& q, Point& p) { F::modcpy(q.x, p.x); F::modcpy(q.y, p.y); F::modcpy(q.z, p.z); }
    static MA_DEV void neg(Point& p) { F::modneg(p.x, p.x); }                        // edwards.c:66-69
    static MA_DEV void inf(Point& p) { F::modzer(p.x); F::modone(p.y); F::modone(p.z); }   // edwards.c:171-176
    static MA_DEV int isinf(const Point& p) { return F::modis0(p.x) & F::modcmp(p.y, p.z); }  // edwards.c:179-183
    static MA_DEV void ran(int r, Point& p) {                                       // edwards.c:55-63
        if (r > 1) { F::modmli(p.x, r, p.x); F::modmli(p.y, r, p.y); F::modmli(p.z, r, p.z); }
    }
    static MA_DEV void cmv(int d, const Point& q, Point& p) {                        // edwards.c:200-205
        F::modcmv(d, q.x, p.x); F::modcmv(d, q.y, p.y); F::modcmv(d, q.z, p.z);
    }

    // P += Q (edwards.c:73-111)
    static MA_DEV void add(const Point& q, Point& p) {
        spint A[N], B[N], Cc[N], D[N], E[N], Ff[N], G[N];
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
        F::modsub(p.x, Cc, p.x);
        F::modsub(p.x, D, p.x);
        F::modmul(p.x, Ff, p.x);
        F::modmul(p.x, A, p.x);
        if constexpr (C::A == -1) F::modadd(D, Cc, p.y); else F::modsub(D, Cc, p.y);
        F::modmul(p.y, A, p.y);
        F::modmul(p.y, G, p.y);
        F::modmul(Ff, G, p.z);
    }
    static MA_DEV void sub(const Point& q, Point& p) {                               // edwards.c:114-119
        Point w;
        cpy(q, w);
        neg(w);
        add(w, p);
    }
    // P = 2P (edwards.c:123-145)
    static MA_DEV void dbl(Point& p) {
        spint B[N], Cc[N], D[N], E[N], Ff[N], H[N], J[N];
        F::modadd(p.x, p.y, B);
        F::modsqr(B, B);
        F::modsqr(p.x, Cc);
        F::modsqr(p.y, D);
        F::modsqr(p.z, H);
        F::modadd(H, H, H);
        if constexpr (C::A == -1) F::modneg(Cc, E); else F::modcpy(Cc, E);
        F::modadd(E, D, Ff);
        F::modsub(Ff, H, J);
        F::modsub(B, Cc, p.x);
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
    // edwards.c:208-218
    static MA_DEV int cmp(const Point& p, const Point& q) {
        spint a[N], b[N];
        F::modmul(p.x, q.z, a);
        F::modmul(q.x, p.z, b);
        int eq = F::modcmp(a, b);
        F::modmul(p.y, q.z, a);
        F::modmul(q.y, p.z, b);
        return eq & F::modcmp(a, b);
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
        static_for<0, N>([&](auto I) { gx[I] = C::gx(I); gy[I] = C::gy(I); });
        setxy<0>(0, gx, gy, p);
    }

    // ---- window table in the global workspace: slot of this lane, entry k
    struct Table {
        spint* base;       // workspace + lane
        size_t stride;     // total lanes
        MA_DEV void put(int k, const Point& w) const {
            static_for<0, N>([&](auto I) {
                base[((size_t)(k * 3 + 0) * N + I) * stride] = w.x[I];
                base[((size_t)(k * 3 + 1) * N + I) * stride] = w.y[I];
                base[((size_t)(k * 3 + 2) * N + I) * stride] = w.z[I];
            });
        }
        MA_DEV void get(int k, Point& w) const {
            static_for<0, N>([&](auto I) {
                w.x[I] = base[((size_t)(k * 3 + 0) * N + I) * stride];
                w.y[I] = base[((size_t)(k * 3 + 1) * N + I) * stride];
                w.z[I] = base[((size_t)(k * 3 + 2) * N + I) * stride];
            });
        }
    };
    static constexpr size_t TABLE_WORDS = 9 * 3 * N;   // per lane

    // constant-time lookup of sign(b) * W[|b|] (edwards.c:381-401): every entry is read
    static MA_DEV void select(int b, const Table& W, Point& p) {
        const int m = b >> 31;
        const int babs = (b ^ m) - m;
#pragma unroll 1
        for (int k = 0; k <= 8; k++) {
            Point w;
            W.get(k, w);
            const int eq = (((babs ^ k) - 1) >> 31) & 1;
            cmv(eq, w, p);
        }
        Point mp;
        cpy(p, mp);
        neg(mp);
        cmv(m & 1, mp, p);
    }

    // P = e*P, signed 4-bit fixed window (edwards.c:435-482).  ew = the scalar as NW little-endian 64-bit
    // words.  Window digits are produced top-down from a left-aligned copy of the scalar and the mask of
    // recoding carries, so no per-lane digit array is needed.
    static MA_DEV void mul(const spint* ew, Point& p, const Table& W) {
        // table W[0..8] = 0, P, 2P, ..., 8P built exactly as edwards.c:441-449 orders it (even entries by
        // doubling W[k/2], odd entries as W[k-1] + P), rolled into one loop so that the instruction stream
        // holds a single copy of dbl and add
        Point Q;
        inf(Q);
        {
            Point T;
            inf(T);
            W.put(0, T);
            W.put(1, p);
#pragma unroll 1
            for (int k = 2; k <= 8; k++) {
                if (k & 1) { W.get(k - 1, T); add(p, T); }
                else       { W.get(k >> 1, T); dbl(T); }
                W.put(k, T);
            }
        }

        // recoding carries: c_0 = 0, c_{j+1} = (nibble_j + c_j > 7)   (edwards.c:461-467)
        spint nib[NW], car[NW];
        static_for<0, NW>([&](auto K) { nib[K] = ew[K]; car[K] = 0; });
        unsigned c = 0;
        static_for<0, NW>([&](auto K) {
            spint word = ew[K], cw = 0;
#pragma unroll 1
            for (int j = 0; j < 16; j++) {
                cw |= (spint)c << j;
                unsigned v = (unsigned)(word & 15) + c;
                c = v > 7 ? 1u : 0u;
                word >>= 4;
            }
            car[K] = cw;              // bit j = carry INTO nibble 16K + j
        });
        // top digit w[2*NB] = final carry
        select((int)c, W, p);
        // iterate nibbles from the top: keep nib left-aligned (top nibble in bits 63..60 of nib[NW-1])
        // and car left-aligned (carry into the current nibble in bit 63 of car[NW-1]); the carry OUT of the
        // current nibble is the carry into the one above, i.e. the bit we consumed in the previous step.
        static_for<0, NW>([&](auto K) { car[K] <<= 48; });   // 16 carry bits per word -> top of the word
        unsigned cout = c;
#pragma unroll 1
        for (int i = 2 * NB - 1; i >= 0; i--) {
            const unsigned nb4 = (unsigned)(nib[NW - 1] >> 60);
            const unsigned cin = (unsigned)(car[NW - 1] >> 63);
            // shift the nibble registers left by 4 and the carry registers so that the next bit is on top
            static_for<0, NW>([&](auto KK) {
                constexpr int k = NW - 1 - KK;
                nib[k] <<= 4;
                if constexpr (k > 0) nib[k] |= nib[k - 1] >> 60;
            });
            // carries: 16 valid bits per word at the top; after consuming 16 of them move to the next word
            car[NW - 1] <<= 1;
            if ((i & 15) == 0) {
                static_for<0, NW - 1>([&](auto KK) {
                    constexpr int k = NW - 1 - KK;
                    car[k] = car[k - 1];
                });
            }
            const int digit = (int)(nb4 + cin) - (int)(cout << 4);
            cout = cin;
            select(digit, W, Q);
#pragma unroll 1
            for (int r = 0; r < 4; r++) dbl(p);
            add(Q, p);
        }
    }

    // R = e*P + f*Q (edwards.c:486-510).  The reference walks a joint sparse form (dnaf, 404-431) with
    // data-dependent branches ("not constant time"); here every lane runs the same 8*NB+7 steps: digit
    // w_i = bit_i(3e) - bit_i(e) + 3*(bit_i(3f) - bit_i(f)) in -4..4, table {O, P, Q-P, Q, Q+P} in the
    // workspace, lookup by full scan + predicated negation, and an unconditional (complete) addition,
    // adding O for a zero digit.  Same point, possibly another projective representative.
    static constexpr int NW1 = NW + 1;
    static MA_DEV void triple(const spint* x, spint* x3) {     // x3 = 3*x over NW+1 words
        spint carry = 0;
        static_for<0, NW>([&](auto K) {
            dpint t = (dpint)x[K] * 3u + carry;
            x3[K] = (spint)t;
            carry = (spint)(t >> 64);
        });
        x3[NW] = carry;
    }
    static MA_DEV void mul2(const spint* ew, const Point& p, const spint* fw, const Point& q, Point& r, const Table& W) {
        {
            Point t;
            inf(t); W.put(0, t);
            W.put(1, p);
            W.put(3, q);
            cpy(q, t); sub(p, t); W.put(2, t);       // Q - P
            cpy(q, t); add(p, t); W.put(4, t);       // Q + P
        }
        // left-aligned copies of e, 3e, f, 3f over NW+1 words; bit 8*NB+7 sits in bit 7 of the top word,
        // so shift everything left by 56 first
        spint e1[NW1], e3[NW1], f1[NW1], f3[NW1];
        static_for<0, NW>([&](auto K) { e1[K] = ew[K]; f1[K] = fw[K]; });
        e1[NW] = 0; f1[NW] = 0;
        triple(ew, e3);
        triple(fw, f3);
        auto shl = [&](spint* v, int sh) {
            static_for<0, NW1>([&](auto KK) {
                constexpr int k = NW1 - 1 - KK;
                v[k] <<= sh;
                if constexpr (k > 0) v[k] |= v[k - 1] >> (64 - sh);
            });
        };
        shl(e1, 56); shl(e3, 56); shl(f1, 56); shl(f3, 56);
        inf(r);
#pragma unroll 1
        for (int i = 8 * NB + 7; i >= 1; i--) {
            const int d = (int)(e3[NW] >> 63) - (int)(e1[NW] >> 63) + 3 * ((int)(f3[NW] >> 63) - (int)(f1[NW] >> 63));
            shl(e1, 1); shl(e3, 1); shl(f1, 1); shl(f3, 1);
            dbl(r);
            const int m = d >> 31;
            const int dabs = (d ^ m) - m;
            Point t, sel;
            inf(sel);
#pragma unroll 1
            for (int k = 0; k <= 4; k++) {
                W.get(k, t);
                cmv((((dabs ^ k) - 1) >> 31) & 1, t, sel);
            }
            cpy(sel, t);
            neg(t);
            cmv(m & 1, t, sel);
            add(sel, r);
        }
    }

    // ---- SoA load / store of a point batch: P[(c*N + i)*ld + j]
    static MA_DEV void load(const spint* Pb, size_t ld, size_t j, Point& p) {
        static_for<0, N>([&](auto I) {
            p.x[I] = Pb[((size_t)(0 * N + I)) * ld + j];
            p.y[I] = Pb[((size_t)(1 * N + I)) * ld + j];
            p.z[I] = Pb[((size_t)(2 * N + I)) * ld + j];
        });
    }
    static MA_DEV void store(spint* Pb, size_t ld, size_t j, const Point& p) {
        static_for<0, N>([&](auto I) {
            Pb[((size_t)(0 * N + I)) * ld + j] = p.x[I];
            Pb[((size_t)(1 * N + I)) * ld + j] = p.y[I];
            Pb[((size_t)(2 * N + I)) * ld + j] = p.z[I];
        });
    }
};

// ---------------------------------------------------------------- kernels
template <class C>
__global__ __launch_bounds__(64) void k_ed_mul(const spint* e, spint* Pb, size_t n, size_t ld, spint* ws) {
    using E = Edwards<C>;
    const size_t lanes = (size_t)gridDim.x * blockDim.x;
    const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    typename E::Table W{ws + lane, lanes};
    for (size_t t = lane; t < n; t += lanes) {
        spint ew[E::NW];
        // big-endian byte record -> little-endian words
        static_for<0, E::NW>([&](auto K) { ew[K] = __builtin_bswap64(e[t * E::NW + (E::NW - 1 - K)]); });
        typename E::Point p;
        E::load(Pb, ld, t, p);
        E::mul(ew, p, W);
        E::store(Pb, ld, t, p);
    }
}

template <class C>
__global__ __launch_bounds__(64) void k_ed_mul2(const spint* e, const spint* Pb, const spint* f, const spint* Qb, spint* Rb,
                                                size_t n, size_t ld, spint* ws) {
    using E = Edwards<C>;
    const size_t lanes = (size_t)gridDim.x * blockDim.x;
    const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    typename E::Table W{ws + lane, lanes};
    for (size_t t = lane; t < n; t += lanes) {
        spint ew[E::NW], fw[E::NW];
        static_for<0, E::NW>([&](auto K) { ew[K] = __builtin_bswap64(e[t * E::NW + (E::NW - 1 - K)]); });
        static_for<0, E::NW>([&](auto K) { fw[K] = __builtin_bswap64(f[t * E::NW + (E::NW - 1 - K)]); });
        typename E::Point p, q, r;
        E::load(Pb, ld, t, p);
        E::load(Qb, ld, t, q);
        E::mul2(ew, p, fw, q, r, W);
        E::store(Rb, ld, t, r);
    }
}

// ecnXXXran: randomise the projective representative by a small factor r (edwards.c:55-63)
template <class C>
__global__ __launch_bounds__(BLOCK) void k_ed_ran(int r, spint* Pb, size_t n, size_t ld) {
    using E = Edwards<C>;
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (size_t)gridDim.x * BLOCK) {
        typename E::Point p;
        E::load(Pb, ld, t, p);
        E::ran(r, p);
        E::store(Pb, ld, t, p);
    }
}

enum { ED_ADD = 0, ED_SUB, ED_DBL, ED_NEG, ED_INF, ED_GEN, ED_COF, ED_AFFINE, ED_CPY };
template <class C, int OP>
__global__ __launch_bounds__(BLOCK) void k_ed_op(const spint* Qb, spint* Pb, size_t n, size_t ld) {
    using E = Edwards<C>;
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (size_t)gridDim.x * BLOCK) {
        typename E::Point p, q;
        if constexpr (OP != ED_INF && OP != ED_GEN && OP != ED_CPY) E::load(Pb, ld, t, p);
        if constexpr (OP == ED_ADD || OP == ED_SUB || OP == ED_CPY) E::load(Qb, ld, t, q);
        if constexpr (OP == ED_ADD) E::add(q, p);
        if constexpr (OP == ED_SUB) E::sub(q, p);
        if constexpr (OP == ED_DBL) E::dbl(p);
        if constexpr (OP == ED_NEG) E::neg(p);
        if constexpr (OP == ED_INF) E::inf(p);
        if constexpr (OP == ED_GEN) E::gen(p);
        if constexpr (OP == ED_COF) E::cof(p);
        if constexpr (OP == ED_AFFINE) E::affine(p);
        if constexpr (OP == ED_CPY) E::cpy(q, p);
        E::store(Pb, ld, t, p);
    }
}

template <class C, bool CMP>
__global__ __launch_bounds__(BLOCK) void k_ed_pred(const spint* Pb, const spint* Qb, int* out, size_t n, size_t ld) {
    using E = Edwards<C>;
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (size_t)gridDim.x * BLOCK) {
        typename E::Point p, q;
        E::load(Pb, ld, t, p);
        if constexpr (CMP) {
            E::load(Qb, ld, t, q);
            out[t] = E::cmp(p, q);
        } else {
            out[t] = E::isinf(p);
        }
    }
}

// ecnXXXset (edwards.c:347-366): big-endian coordinate records x and/or y (either may be null), s = sign array or null
template <class C, int MODE>
__global__ __launch_bounds__(BLOCK) void k_ed_set(const int* s, const spint* xb, const spint* yb, spint* Pb, size_t n, size_t ld) {
    using E = Edwards<C>;
    using F = typename E::F;
    constexpr int NW = E::NW;
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (size_t)gridDim.x * BLOCK) {
        spint X[E::N], Y[E::N], w[NW];
        if constexpr (MODE != 2) {
            static_for<0, NW>([&](auto K) { w[K] = __builtin_bswap64(xb[t * NW + (NW - 1 - K)]); });
            (void)F::modimp_words(w, X);
        }
        if constexpr (MODE != 1) {
            static_for<0, NW>([&](auto K) { w[K] = __builtin_bswap64(yb[t * NW + (NW - 1 - K)]); });
            (void)F::modimp_words(w, Y);
        }
        typename E::Point p;
        E::template setxy<MODE>(s ? s[t] : 0, X, Y, p);
        E::store(Pb, ld, t, p);
    }
}

// ecnXXXget (edwards.c:221-239): makes P affine (written back), exports x and/or y, sign of the omitted coordinate
template <class C>
__global__ __launch_bounds__(BLOCK) void k_ed_get(spint* Pb, spint* xb, spint* yb, int* sign, size_t n, size_t ld) {
    using E = Edwards<C>;
    using F = typename E::F;
    constexpr int NW = E::NW;
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (size_t)gridDim.x * BLOCK) {
        typename E::Point p;
        E::load(Pb, ld, t, p);
        E::affine(p);
        E::store(Pb, ld, t, p);
        spint w[NW];
        if (xb) {
            F::modexp_words(p.x, w);
            static_for<0, NW>([&](auto K) { xb[t * NW + (NW - 1 - K)] = __builtin_bswap64(w[K]); });
        }
        if (yb) {
            F::modexp_words(p.y, w);
            static_for<0, NW>([&](auto K) { yb[t * NW + (NW - 1 - K)] = __builtin_bswap64(w[K]); });
        }
        if (sign) {
            int sg = 0;
            if (!yb) sg = F::modsign(p.y);
            else if (!xb) sg = F::modsign(p.x);
            sign[t] = sg;
        }
    }
}

}  // namespace ma
