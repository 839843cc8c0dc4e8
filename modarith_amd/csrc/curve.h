// modarith_amd/csrc/curve.h -- the part of the reference's curve layer that is the same for edwards.c and
// weierstrass.c (cmv, cmp, ran, sub, select, the signed 4-bit fixed-window mul, mul2), written once over a
// curve class K that supplies the formulas (add, dbl, neg, inf, isinf, affine, setxy, gen, cof), plus the
// batched kernels of the curve API (curve.h:13-29).  One point per lane, coordinates in VGPRs, built from the
// bit-exact Field<P> functions in the reference's order.  The 9-entry window table of ecnXXXmul does not fit the
// register file (9 x 3 x N limbs); it lives in a per-lane slot of a global workspace laid out
// [entry][coord][limb][lane] (coalesced; sized to the resident grid so it stays in the 256 MiB Infinity Cache) and
// is scanned in full on every lookup: the table index never forms an address, selection is lane-predicated modcmv.
#pragma once
#include "field.h"
#include "kernels.h"

namespace ma {

template <class Crv, class P_>
struct CurveOps {
    using P = P_;
    using F = Field<P, true>;   // FAST product path where the driver proved it (P::SPLIT > 0), else exact
    static constexpr int N = P::N;
    static constexpr int NB = P::NBYTES;
    static constexpr int NW = (NB + 7) / 8;          // 64-bit words of a scalar / coordinate record
    static constexpr int PADB = 8 * NW - NB;         // unused top bytes of the top word (6 for the 66-byte NIST521 records)
    // v <<= S over W words, S a compile-time bit count
    template <int S, int W>
    static MA_DEV void shl_words(spint* v) {
        constexpr int ws = S / 64, bs = S % 64;
        static_for<0, W>([&](auto KK) {
            constexpr int k = W - 1 - KK;
            spint x = 0;
            if constexpr (k - ws >= 0) x = v[k - ws] << bs;
            if constexpr (bs != 0 && k - ws - 1 >= 0) x |= v[k - ws - 1] >> (64 - bs);
            v[k] = x;
        });
    }
    struct Point { spint x[N], y[N], z[N]; };

    static MA_DEV void cpy(const Point& q, Point& p) { F::modcpy(q.x, p.x); F::modcpy(q.y, p.y); F::modcpy(q.z, p.z); }
    static MA_DEV void ran(int r, Point& p) {                                       // edwards.c:55-63
        if (r > 1) { F::modmli(p.x, r, p.x); F::modmli(p.y, r, p.y); F::modmli(p.z, r, p.z); }
    }
    static MA_DEV void cmv(int d, const Point& q, Point& p) {                        // edwards.c:200-205
        F::modcmv(d, q.x, p.x); F::modcmv(d, q.y, p.y); F::modcmv(d, q.z, p.z);
    }
    static MA_DEV void sub(const Point& q, Point& p) {                               // edwards.c:114-119
        Point w;
        cpy(q, w);
        Crv::neg(w);
        Crv::add(w, p);
    }
    // edwards.c:208-218 / weierstrass.c:320-330
    static MA_DEV int cmp(const Point& p, const Point& q) {
        spint a[N], b[N];
        F::modmul(p.x, q.z, a);
        F::modmul(q.x, p.z, b);
        int eq = F::modcmp(a, b);
        F::modmul(p.y, q.z, a);
        F::modmul(q.y, p.z, b);
        return eq & F::modcmp(a, b);
    }

    // ---- window tables in the global workspace: slot of this lane, entry k (mul: one 9-entry table; mul2: two)
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
    static constexpr size_t TABLE_WORDS = 2 * 9 * 3 * N;   // per lane: room for the two tables of mul2

    // constant-time lookup of sign(b) * W[|b|] (edwards.c:381-401): every entry is read
    static MA_DEV void select(int b, const Table& W, Point& p) {
        const int m = b >> 31;
        const int babs = (b ^ m) - m;
        // W[0] is the neutral element as inf() writes it: start from it and scan entries 1..8 only (+2 % on ED25519).  Not
        // for the 9-limb Weierstrass kernel (NIST P-521): there this form makes the register allocator spill 2 928 instead
        // of 652 VGPRs, 7 000 scratch accesses per window, and the kernel runs 8x slower (4.3e5 instead of 3.3e6 per s) --
        // it keeps the nine-entry scan.
        constexpr int K0 = Crv::SELECT_FROM_NEUTRAL ? 1 : 0;
        if constexpr (Crv::SELECT_FROM_NEUTRAL) Crv::inf(p);
#pragma unroll 1    // rolled: unrolling 3x / 9x measured -2 % / -17 % (more live loads, same latency chain)
        for (int k = K0; k <= 8; k++) {
            Point w;
            W.get(k, w);
            const int eq = (((babs ^ k) - 1) >> 31) & 1;
            cmv(eq, w, p);
        }
        Point mp;
        cpy(p, mp);
        Crv::neg(mp);
        cmv(m & 1, mp, p);
    }

    // table W[0..8] = 0, P, 2P, ..., 8P built exactly as edwards.c:441-449 orders it (even entries by doubling
    // W[k/2], odd entries as W[k-1] + P), rolled into one loop so that the instruction stream holds a single
    // copy of dbl and add
    static MA_DEV void build_table(const Point& p, const Table& W) {
        Point T;
        Crv::inf(T);
        W.put(0, T);
        W.put(1, p);
#pragma unroll 1
        for (int k = 2; k <= 8; k++) {
            if (k & 1) { W.get(k - 1, T); Crv::add(p, T); }
            else       { W.get(k >> 1, T); Crv::dbl(T); }
            W.put(k, T);
        }
    }

    // Signed 4-bit recoding of a scalar (edwards.c:452-467), produced digit by digit from the top: the scalar stays
    // left-aligned in NW words (nib), the carries c_0 = 0, c_{j+1} = (nibble_j + c_j > 7) are computed once into a bit
    // mask (car), so no per-lane digit array is needed.  top() is the digit w[2*NB] (the final carry), next() then
    // returns w[2*NB-1], ..., w[0].
    struct Recoder {
        spint nib[NW], car[NW];
        unsigned cout;
        int consumed;
        MA_DEV int top(const spint* ew) {
            // left-aligned (a no-op shift when Nbytes is a multiple of 8): the padding nibbles at the bottom are zero,
            // produce no carry and are never reached by the 2*NB digits
            static_for<0, NW>([&](auto K) { nib[K] = ew[K]; car[K] = 0; });
            shl_words<8 * PADB, NW>(nib);
            unsigned c = 0;
            static_for<0, NW>([&](auto K) {
                spint word = nib[K], cw = 0;
#pragma unroll 1
                for (int j = 0; j < 16; j++) {
                    cw |= (spint)c << j;
                    unsigned v = (unsigned)(word & 15) + c;
                    c = v > 7 ? 1u : 0u;
                    word >>= 4;
                }
                car[K] = cw;              // bit j = carry INTO nibble 16K + j
            });
            // car left-aligned: the carry into the current nibble in bit 63 of car[NW-1]; the carry OUT of the current
            // nibble is the carry into the one above, i.e. the bit consumed in the previous step
            static_for<0, NW>([&](auto K) { car[K] <<= 48; });   // 16 carry bits per word -> top of the word
            cout = c;
            consumed = 0;
            return (int)c;
        }
        MA_DEV int next() {
            const unsigned nb4 = (unsigned)(nib[NW - 1] >> 60);
            const unsigned cin = (unsigned)(car[NW - 1] >> 63);
            static_for<0, NW>([&](auto KK) {
                constexpr int k = NW - 1 - KK;
                nib[k] <<= 4;
                if constexpr (k > 0) nib[k] |= nib[k - 1] >> 60;
            });
            car[NW - 1] <<= 1;
            consumed++;
            if ((consumed & 15) == 0) {          // 16 digits consumed: the next carry word moves up
                static_for<0, NW - 1>([&](auto KK) {
                    constexpr int k = NW - 1 - KK;
                    car[k] = car[k - 1];
                });
            }
            const int digit = (int)(nb4 + cin) - (int)(cout << 4);
            cout = cin;
            return digit;
        }
    };

    // P = e*P, signed 4-bit fixed window (edwards.c:435-482).  ew = the scalar as NW little-endian 64-bit words.
    static MA_DEV void mul(const spint* ew, Point& p, const Table& W) {
        Point Q;
        Crv::inf(Q);
        build_table(p, W);
        Recoder rc;
        select(rc.top(ew), W, p);
#pragma unroll 1
        for (int i = 2 * NB - 1; i >= 0; i--) {
            select(rc.next(), W, Q);
#pragma unroll 1
            for (int r = 0; r < 4; r++) Crv::dbl(p);
            Crv::add(Q, p);
        }
    }

    // R = e*P + f*Q (edwards.c:486-510).  The reference walks a joint sparse form (dnaf, 404-431) with data-dependent
    // branches ("not constant time"), which would diverge across lanes.  Here: two signed 4-bit fixed-window
    // multiplications sharing their doublings -- tables {0..8}P and {0..8}Q in the workspace, per window four
    // doublings and two complete additions -- so every lane runs the same 2*NB windows and the multiplication is
    // constant-time as well.  Same point as the reference's, another projective representative.
    static MA_DEV void mul2(const spint* ew, const Point& p, const spint* fw, const Point& q, Point& r, const Table& W) {
        // the two halves (table, lookup, addition) run through loops of two rolled iterations, so the instruction
        // stream holds one copy of add / dbl / select, as in mul; t is wave-uniform
        const size_t second = (size_t)9 * 3 * N * W.stride;
        Point T;
#pragma unroll 1
        for (int t = 0; t < 2; t++) {
            cpy(p, T);
            if (t) cpy(q, T);
            build_table(T, Table{W.base + (size_t)t * second, W.stride});
        }
        Recoder re, rf;
        const int top_e = re.top(ew), top_f = rf.top(fw);
        Crv::inf(r);
#pragma unroll 1
        for (int t = 0; t < 2; t++) {
            select(t ? top_f : top_e, Table{W.base + (size_t)t * second, W.stride}, T);
            Crv::add(T, r);                      // first pass adds to the neutral element
        }
#pragma unroll 1
        for (int i = 2 * NB - 1; i >= 0; i--) {
#pragma unroll 1
            for (int k = 0; k < 4; k++) Crv::dbl(r);
            const int de = re.next(), df = rf.next();
#pragma unroll 1
            for (int t = 0; t < 2; t++) {
                select(t ? df : de, Table{W.base + (size_t)t * second, W.stride}, T);
                Crv::add(T, r);
            }
        }
    }

    // R = e*P + f*Q with the REFERENCE'S OWN walk (edwards.c:404-431 dnaf, 486-510; weierstrass.c:545-569): the joint sparse form
    // w[k] = (bit_k(3e) - bit_k(e)) + 3 (bit_k(3f) - bit_k(f)) in {-4..4}, table W = {O, P, Q-P, Q, Q+P}, R = O, and from the first
    // non-zero digit down to k = 1: R = 2R, then R += W[w] or R -= W[-w].  Same field calls in the same order, hence the reference's
    // projective limbs -- and, like the reference ("not constant time"), a walk that depends on the scalars: lanes of a wave start
    // at different digits and skip different additions, so the wave pays for the union of their paths (measured: 4-18 % slower than mul2 above, tools/time_mul2.py).
    // The digits are produced from the top by shifting e, 3e, f, 3f left one bit per step; no digit array is stored.
    static MA_DEV void mul2_exact(const spint* ew, const Point& p, const spint* fw, const Point& q, Point& r, const Table& W) {
        constexpr int NX = NW + 1;                        // 3e needs two more bits than e
        constexpr int TOP = 8 * NB + 7;                   // index of the highest digit (edwards.c:497)
        Point T;
        Crv::inf(T); W.put(0, T);
        W.put(1, p);
        W.put(3, q);
        cpy(q, T); sub(p, T); W.put(2, T);                // Q - P
        cpy(q, T); Crv::add(p, T); W.put(4, T);           // Q + P
        spint a[NX], a3[NX], b[NX], b3[NX];
        static_for<0, NX>([&](auto K) { a[K] = K < NW ? ew[K < NW ? K : 0] : 0; b[K] = K < NW ? fw[K < NW ? K : 0] : 0; });
        {   // 3x = x + 2x over NX words
            spint ca = 0, cb = 0;
            static_for<0, NX>([&](auto K) {
                const spint ta = (a[K] << 1) | (K > 0 ? a[K > 0 ? K - 1 : 0] >> 63 : 0);
                const spint tb = (b[K] << 1) | (K > 0 ? b[K > 0 ? K - 1 : 0] >> 63 : 0);
                const spint sa = a[K] + ta, sa2 = sa + ca;
                ca = (spint)(sa < ta) | (spint)(sa2 < sa);
                a3[K] = sa2;
                const spint sb = b[K] + tb, sb2 = sb + cb;
                cb = (spint)(sb < tb) | (spint)(sb2 < sb);
                b3[K] = sb2;
            });
        }
        constexpr int S = NX * 64 - 1 - TOP;              // left-align: digit TOP in bit 63 of the top word
        shl_words<S, NX>(a); shl_words<S, NX>(a3); shl_words<S, NX>(b); shl_words<S, NX>(b3);
        Crv::inf(r);
        bool started = false;
#pragma unroll 1
        for (int i = TOP; i >= 1; i--) {
            const int j = ((int)(a3[NX - 1] >> 63) - (int)(a[NX - 1] >> 63)) + 3 * ((int)(b3[NX - 1] >> 63) - (int)(b[NX - 1] >> 63));
            shl_words<1, NX>(a); shl_words<1, NX>(a3); shl_words<1, NX>(b); shl_words<1, NX>(b3);
            if (!started) {
                if (j == 0) continue;                     // "ignore leading zeros" (edwards.c:498)
                started = true;
            }
            Crv::dbl(r);
            if (j != 0) {
                W.get(j < 0 ? -j : j, T);
                if (j < 0) Crv::neg(T);                   // ecnXXXsub = copy, negate, add (edwards.c:114-119)
                Crv::add(T, r);
            }
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
// resident waves per SIMD the scalar-multiplication kernels are register-budgeted for (256 VGPRs each); three waves at
// 170 VGPRs measured the same throughput, one wave at 512 VGPRs 20 % less
#define MA_MUL_WPS 2
template <class Crv>
__global__ __launch_bounds__(64, MA_MUL_WPS) void k_ed_mul(const unsigned char* e, spint* Pb, size_t n, size_t ld, spint* ws) {
    using E = Crv;
    const size_t lanes = (size_t)gridDim.x * blockDim.x;
    const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    typename E::Table W{ws + lane, lanes};
    for (size_t t = lane; t < n; t += lanes) {
        spint ew[E::NW];
        load_be_record<typename E::P>(e, t, ew);         // big-endian byte record -> little-endian words
        typename E::Point p;
        E::load(Pb, ld, t, p);
        E::mul(ew, p, W);
        E::store(Pb, ld, t, p);
    }
}

template <class Crv>
__global__ __launch_bounds__(64, MA_MUL_WPS) void k_ed_mul2(const unsigned char* e, const spint* Pb, const unsigned char* f, const spint* Qb, spint* Rb,
                                                size_t n, size_t ld, spint* ws) {
    using E = Crv;
    const size_t lanes = (size_t)gridDim.x * blockDim.x;
    const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    typename E::Table W{ws + lane, lanes};
    for (size_t t = lane; t < n; t += lanes) {
        spint ew[E::NW], fw[E::NW];
        load_be_record<typename E::P>(e, t, ew);
        load_be_record<typename E::P>(f, t, fw);
        typename E::Point p, q, r;
        E::load(Pb, ld, t, p);
        E::load(Qb, ld, t, q);
        E::mul2(ew, p, fw, q, r, W);
        E::store(Rb, ld, t, r);
    }
}

template <class Crv>
__global__ __launch_bounds__(64, MA_MUL_WPS) void k_ed_mul2x(const unsigned char* e, const spint* Pb, const unsigned char* f, const spint* Qb, spint* Rb,
                                                 size_t n, size_t ld, spint* ws) {
    using E = Crv;
    const size_t lanes = (size_t)gridDim.x * blockDim.x;
    const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    typename E::Table W{ws + lane, lanes};
    for (size_t t = lane; t < n; t += lanes) {
        spint ew[E::NW], fw[E::NW];
        load_be_record<typename E::P>(e, t, ew);
        load_be_record<typename E::P>(f, t, fw);
        typename E::Point p, q, r;
        E::load(Pb, ld, t, p);
        E::load(Qb, ld, t, q);
        E::mul2_exact(ew, p, fw, q, r, W);
        E::store(Rb, ld, t, r);
    }
}

// ecnXXXran: randomise the projective representative by a small factor r (edwards.c:55-63)
template <class Crv>
__global__ __launch_bounds__(BLOCK) void k_ed_ran(int r, spint* Pb, size_t n, size_t ld) {
    using E = Crv;
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (size_t)gridDim.x * BLOCK) {
        typename E::Point p;
        E::load(Pb, ld, t, p);
        E::ran(r, p);
        E::store(Pb, ld, t, p);
    }
}

enum { ED_ADD = 0, ED_SUB, ED_DBL, ED_NEG, ED_INF, ED_GEN, ED_COF, ED_AFFINE, ED_CPY };
template <class Crv, int OP>
__global__ __launch_bounds__(BLOCK) void k_ed_op(const spint* Qb, spint* Pb, size_t n, size_t ld) {
    using E = Crv;
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

template <class Crv, bool CMP>
__global__ __launch_bounds__(BLOCK) void k_ed_pred(const spint* Pb, const spint* Qb, int* out, size_t n, size_t ld) {
    using E = Crv;
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
template <class Crv, int MODE>
__global__ __launch_bounds__(BLOCK) void k_ed_set(const int* s, const unsigned char* xb, const unsigned char* yb, spint* Pb, size_t n, size_t ld) {
    using E = Crv;
    using F = typename E::F;
    constexpr int NW = E::NW;
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (size_t)gridDim.x * BLOCK) {
        spint X[E::N], Y[E::N], w[NW];
        if constexpr (MODE != 2) {
            load_be_record<typename E::P>(xb, t, w);
            (void)F::modimp_words(w, X);
        }
        if constexpr (MODE != 1) {
            load_be_record<typename E::P>(yb, t, w);
            (void)F::modimp_words(w, Y);
        }
        typename E::Point p;
        E::template setxy<MODE>(s ? s[t] : 0, X, Y, p);
        E::store(Pb, ld, t, p);
    }
}

// ecnXXXget (edwards.c:221-239): makes P affine (written back), exports x and/or y, sign of the omitted coordinate
template <class Crv>
__global__ __launch_bounds__(BLOCK) void k_ed_get(spint* Pb, unsigned char* xb, unsigned char* yb, int* sign, size_t n, size_t ld) {
    using E = Crv;
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
            store_be_record<typename E::P>(xb, t, w);
        }
        if (yb) {
            F::modexp_words(p.y, w);
            store_be_record<typename E::P>(yb, t, w);
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
