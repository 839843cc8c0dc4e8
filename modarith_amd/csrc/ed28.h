// modarith_amd/csrc/ed28.h -- fused ecnXXXmul + ecnXXXget for ED448 on the fe28 representation (gfx950).
//
// Same construction as ed26.h (which see): only canonical affine bytes leave the kernel, so the arithmetic runs on
// sixteen 28-bit limbs (fe28.h: golden-ratio Karatsuba products) and on extended coordinates (X:Y:Z:T) with the
// Hisil-Wong-Carter-Dawson formulas for a = 1, which are complete on x^2 + y^2 = 1 - 39081 x^2 y^2 (a = 1 is a square,
// d = -39081 a non-square mod 2^448 - 2^224 - 1): the affine point the reference's edwards.c formulas reach, for every
// input point on the curve.  The coordinates arrive in Montgomery form (monty.py, R = 2^504): a projective point whose
// three coordinates carry the same factor R is the same point, so the limbs are simply read as integers mod p.
//
// 3-bit signed fixed windows like ed26.h: e' = e + sum_{i<150} 4*8^i < 8^150, digit_i = window_i(e') - 4.  The table
// {1,2,3,4}P -- affine, cached as canonical packed (x, y, 39081 x y), 4 x 3 x 56 bytes per lane -- does not fit the
// register file next to a 64-register point, so it lives in a caller-provided device workspace laid out
// [entry][word][lane] (8-byte coalesced accesses; 88 MB for the resident grid, inside the Infinity Cache); every lookup
// reads all four entries and selects with v_cndmask.
//
// Limb bounds (fe28.h): "tight" = below 2^28 (+2^9 on limbs 1, 9); add() of tight values stays below 2^29; sub() carries
// back to tight; mul_k / sqr_k (Karatsuba) want one tight operand and one below 2^29 / a tight operand; mul / sqr take
// anything below 2^29 (one operand up to 1.5 * 2^29: a folded column holds 38 products, 57 * 2^58 < 2^64).
#pragma once
#include "fe28.h"

namespace ma {

struct Ed28 {
    using F = Fe28;
    static constexpr uint32_t M28 = F::M28;
    static constexpr uint32_t D_ABS = 39081;       // d = -39081 (curve.py: CONSTANT_B of ED448)
    struct Ext { uint32_t X[16], Y[16], Z[16], T[16]; };

    // limbs below 2^31 -> tight, 32-bit registers only (the carry out of limb 15 re-enters at limbs 0 and 8)
    static MA_DEV void wc(uint32_t* f) {
        static_for<0, 15>([&](auto I) {
            constexpr int i = I;
            f[i + 1] += f[i] >> 28;
            f[i] &= M28;
        });
        const uint32_t top = f[15] >> 28;
        f[15] &= M28;
        const uint32_t h0 = f[0] + top, h8 = f[8] + top;
        f[0] = h0 & M28;
        f[1] += h0 >> 28;
        f[8] = h8 & M28;
        f[9] += h8 >> 28;
    }
    // r = 2p - f for tight f: below 2^29, not carried (limbs of 2p: 2^29-2, limb 8: 2^29-4)
    static MA_DEV void neg2p(const uint32_t* f, uint32_t* r) {
        static_for<0, 16>([&](auto I) {
            constexpr int i = I;
            constexpr uint32_t twop = (i == 8) ? 0x1ffffffcu : 0x1ffffffeu;
            r[i] = twop - f[i];
        });
    }
    // 8 x 56-bit limbs (field.c form, limbs below 2^58: the contract of the curve layer) -> tight fe28
    static MA_DEV void from56(const spint* x, uint32_t* r) {
        static_for<0, 8>([&](auto K) {
            constexpr int k = K;
            r[2 * k] = (uint32_t)x[k] & M28;
            r[2 * k + 1] = (uint32_t)(x[k] >> 28);
        });
        wc(r);
    }

    // P = 2P, dbl-2008-hwcd with a = 1: A = X^2, B = Y^2, C = 2Z^2, G = A + B, E = (X+Y)^2 - G, F = G - C, H = A - B;
    // X3 = E F, Y3 = G H, Z3 = F G, T3 = E H.  Input X, Y, Z tight; T is not read.
    template <bool WANT_T>
    static MA_DEV void dbl(Ext& p) { dbl(p, WANT_T); }
    // want_t is wave-uniform (a loop counter, never data): the window loop keeps ONE copy of the doubling in the
    // instruction stream -- three unrolled copies plus the addition are about 80 KB of code, more than the 64 KB
    // instruction cache, and the kernel then waits on instruction fetch for 40 % of its time (SQ_WAIT_ANY)
    static MA_DEV void dbl(Ext& p, bool want_t) {
        uint32_t A[16], B[16], Cc[16], S[16], G[16], E[16], Ff[16], H[16];
        F::sqr_k(p.X, A);
        F::sqr_k(p.Y, B);
        F::sqr_k(p.Z, Cc);
        F::add(p.X, p.Y, S);        // < 2^29
        F::sqr(S, S);
        F::add(A, B, G);            // < 2^29
        F::sub(S, G, E);            // tight
        F::add(Cc, Cc, Cc);         // < 2^29
        F::sub(G, Cc, Ff);          // tight
        F::sub(A, B, H);            // tight
        F::mul_k(E, Ff, p.X);
        F::mul_k(H, G, p.Y);
        F::mul_k(Ff, G, p.Z);
        if (want_t) F::mul_k(E, H, p.T);
    }
    // common tail of the additions (a = 1, d = -39081): with A = X1 X2, B = Y1 Y2, Cc = 39081 T1 T2 (= -C), D = Z1 Z2,
    // M = (X1+Y1)(X2+Y2):  E = M - A - B, F = D - C = D + Cc, G = D + C = D - Cc, H = B - A;  X3 = E F, Y3 = G H, Z3 = F G
    static MA_DEV void add_tail(const uint32_t* A, const uint32_t* B, const uint32_t* Cc, const uint32_t* D, const uint32_t* M, Ext& p,
                                bool want_t = false) {
        uint32_t E[16], Ff[16], G[16], H[16], AB[16];
        F::add(A, B, AB);           // < 2^29
        F::sub(M, AB, E);           // tight
        F::add(D, Cc, Ff);          // < 2^29
        F::sub(D, Cc, G);           // tight
        F::sub(B, A, H);            // tight
        F::mul_k(E, Ff, p.X);
        F::mul_k(G, H, p.Y);
        F::mul_k(G, Ff, p.Z);
        if (want_t) F::mul_k(E, H, p.T);    // (wave-uniform flag: one copy of the addition serves both uses)
    }
    // P += Q, Q affine and cached as (x, y, td = 39081 x y), already sign-adjusted: xs, tds below 2^29, y tight.
    // Reads T of P; T of the sum is not produced (a doubling follows).
    static MA_DEV void add_cached(Ext& p, const uint32_t* xs, const uint32_t* y, const uint32_t* tds, bool want_t = false) {
        uint32_t A[16], B[16], Cc[16], M[16], s1[16], s2[16];
        F::mul_k(p.X, xs, A);
        F::mul_k(p.Y, y, B);
        F::mul_k(p.T, tds, Cc);
        F::add(p.X, p.Y, s1);       // < 2^29
        F::add(xs, y, s2);          // < 1.5 * 2^29
        F::mul(s1, s2, M);
        add_tail(A, B, Cc, p.Z, M, p, want_t);
    }
    // P += Q, both extended; used once, to build 3P
    static MA_DEV void add_ext(Ext& p, const Ext& q) {
        uint32_t A[16], B[16], Cc[16], D[16], M[16], s1[16], s2[16];
        F::mul_k(p.X, q.X, A);
        F::mul_k(p.Y, q.Y, B);
        F::mul_k(p.T, q.T, Cc);
        F::mul_small<D_ABS>(Cc, Cc);
        F::mul_k(p.Z, q.Z, D);
        F::add(p.X, p.Y, s1);
        F::add(q.X, q.Y, s2);
        F::mul(s1, s2, M);
        add_tail(A, B, Cc, D, M, p);
    }
    // The same with q fetched coordinate by coordinate -- fetch(c, out), c = 0..3 for X, Y, Z, T -- so that the table builder holds
    // one point while it forms 3P (q waits in its table slot)
    template <class FETCH>
    static MA_DEV void add_ext_fetched(Ext& p, FETCH fetch) {
        uint32_t A[16], B[16], Cc[16], D[16], M[16];
        {
            uint32_t q[16];
            fetch(3, q);
            F::mul_k(p.T, q, Cc);
            F::mul_small<D_ABS>(Cc, Cc);
            fetch(2, q);
            F::mul_k(p.Z, q, D);
        }
        {
            uint32_t qx[16], qy[16], s1[16], s2[16];
            fetch(0, qx);
            fetch(1, qy);
            F::mul_k(p.X, qx, A);
            F::mul_k(p.Y, qy, B);
            F::add(p.X, p.Y, s1);
            F::add(qx, qy, s2);
            F::mul(s1, s2, M);
        }
        add_tail(A, B, Cc, D, M, p);
    }
};

constexpr int ED448_TABLE_WORDS = 4 * 3 * 7;       // 64-bit words per lane in the workspace

// ---- where the window table of a lane lives, and where the recoded scalar comes from (round 4).  The functions below are written
// over two small concepts so that the kernels can keep both OUT of the register file while the host check keeps plain arrays:
//   TAB: origin() = pointer to word 0 of this lane's table (a fresh value per call in the slab form, so that row addresses are
//        formed at the access instead of being carried -- and spilled -- across the window), stride() = words between table words;
//   DIG: window(i) = the i-th window of the recoded scalar, in the order the loop consumes them.
struct TabStrided {                         // word k at tab[k * tstride] (host check: tstride = 1; the round-3 kernels: lanes + skew)
    uint64_t* tab;
    size_t tstride;
    MA_DEV uint64_t* origin() const { return tab; }
    MA_DEV size_t stride() const { return tstride; }
};
struct TabSlab {                            // per-wave slab [word][64 lanes]: every access of a wave is one contiguous 512-byte row
    uint64_t* base;                         // wave-uniform
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
// e' = e + sum_{i<150} 4*8^i (450 bits), 150 windows of 3 bits from the top; window(i) must be called for i = 0, 1, 2, ... in order
struct Win3Regs {
    uint64_t w[8];
    MA_DEV void init(const uint64_t* ew) {
        constexpr auto cw = [](int k) {
            uint64_t v = 0;
            for (int b = 0; b < 64; b++) {
                const int pos = 64 * k + b;
                if (pos < 450 && pos % 3 == 2) v |= (uint64_t)1 << b;
            }
            return v;
        };
        unsigned __int128 acc = 0;
        uint64_t s[8];
        static_for<0, 8>([&](auto K) {
            constexpr int k = K;
            acc += (unsigned __int128)(k < 7 ? ew[k < 7 ? k : 0] : 0) + cw(k);
            s[k] = (uint64_t)acc;
            acc >>= 64;
        });
        static_for<0, 8>([&](auto KK) {
            constexpr int k = 7 - KK;
            w[k] = s[k] << 62;
            if constexpr (k > 0) w[k] |= s[k - 1] >> 2;
        });
    }
    MA_DEV uint32_t window(int) {
        const uint32_t win = (uint32_t)(w[7] >> 61);
        static_for<0, 8>([&](auto KK) {
            constexpr int k = 7 - KK;
            w[k] <<= 3;
            if constexpr (k > 0) w[k] |= w[k - 1] >> 61;
        });
        return win;
    }
};
// the same windows, produced once into the lane's column of an LDS array (one byte per window) before the point is loaded
struct Win3Lds {
    const unsigned char* col;               // digs + lane, windows 64 bytes apart
    static MA_DEV void fill(const uint64_t* ew, unsigned char* col) {
        Win3Regs r;
        r.init(ew);
#pragma unroll 1
        for (int i = 0; i < 150; i++) col[(size_t)i * 64] = (unsigned char)r.window(i);
    }
    MA_DEV uint32_t window(int i) const { return col[(size_t)i * 64]; }
};
// e' = e + sum_{i<225} 2*4^i (450 bits), 225 windows of 2 bits from the top (the double multiplication)
struct Win2Regs {
    uint64_t w[8];
    MA_DEV void init(const uint64_t* in) {
        constexpr auto cw = [](int k) {
            uint64_t v = 0;
            for (int b = 0; b < 64; b++) {
                const int pos = 64 * k + b;
                if (pos < 450 && pos % 2 == 1) v |= (uint64_t)1 << b;
            }
            return v;
        };
        unsigned __int128 acc = 0;
        uint64_t s[8];
        static_for<0, 8>([&](auto K) {
            constexpr int k = K;
            acc += (unsigned __int128)(k < 7 ? in[k < 7 ? k : 0] : 0) + cw(k);
            s[k] = (uint64_t)acc;
            acc >>= 64;
        });
        static_for<0, 8>([&](auto KK) {
            constexpr int k = 7 - KK;
            w[k] = s[k] << 62;
            if constexpr (k > 0) w[k] |= s[k - 1] >> 2;
        });
    }
    MA_DEV uint32_t window(int) {
        const uint32_t win = (uint32_t)(w[7] >> 62);
        static_for<0, 8>([&](auto KK) {
            constexpr int k = 7 - KK;
            w[k] <<= 2;
            if constexpr (k > 0) w[k] |= w[k - 1] >> 62;
        });
        return win;
    }
};
struct Win2Lds {                            // four windows per byte (57 bytes per scalar and lane)
    const unsigned char* col;
    static MA_DEV void fill(const uint64_t* in, unsigned char* col) {
        Win2Regs r;
        r.init(in);
#pragma unroll 1
        for (int q = 0; q < 57; q++) {
            unsigned b = 0;
#pragma unroll
            for (int h = 0; h < 4; h++) b |= (4 * q + h < 225 ? r.window(0) : 0u) << (2 * h);
            col[(size_t)q * 64] = (unsigned char)b;
        }
    }
    MA_DEV uint32_t window(int i) const { return ((uint32_t)col[(size_t)(i >> 2) * 64] >> (2 * (i & 3))) & 3u; }
};

// The four entry slots of one lane's table while it is being built (round 4).  Each multiple is STASHED in its slot as canonical
// (X, Y, Z) -- 3 x 7 words, exactly the size of an entry -- as soon as it exists, so that the builder never holds more than two
// points; to_affine_entries() then forms the products of the four Z from the slots (a = Z1 Z2, b = a Z3, c = b Z4), inverts once,
// and overwrites every slot with its affine entry (x, y, 39081 x y).
template <class TAB>
struct Ed448Slots {
    using E = Ed28;
    using F = Fe28;
    const TAB& T;
    MA_DEV void put7(int word, const uint32_t* f) const {
        uint64_t w[7];
        F::to_words(f, w);
        uint64_t* tab = T.origin();
        const size_t tstride = T.stride();
        static_for<0, 7>([&](auto K) { tab[(size_t)(word + K) * tstride] = w[K]; });
    }
    MA_DEV void get7(int word, uint32_t* f) const {
        uint64_t w[7];
        const uint64_t* tab = T.origin();
        const size_t tstride = T.stride();
        static_for<0, 7>([&](auto K) { w[K] = tab[(size_t)(word + K) * tstride]; });
        F::from_words(w, f);
    }
    MA_DEV void stash(const E::Ext& p, int entry) const {
        put7(entry * 21, p.X);
        put7(entry * 21 + 7, p.Y);
        put7(entry * 21 + 14, p.Z);
    }
    MA_DEV void finish(int entry, const uint32_t* zi) const {
        uint32_t x[16], y[16], s[16];
        get7(entry * 21, x);
        get7(entry * 21 + 7, y);
        F::mul_k(x, zi, x);
        F::mul_k(y, zi, y);
        put7(entry * 21, x);
        put7(entry * 21 + 7, y);
        F::mul_k(x, y, s);
        F::mul_small<E::D_ABS>(s, s);
        put7(entry * 21 + 14, s);
    }
    MA_DEV void to_affine_entries() const {
        // Montgomery's trick on the four stashed Z: a = Z1 Z2, b = a Z3, c = b Z4, one inversion, then back down.  Nothing but the
        // product is live across the inversion (its own working set is four elements): a and b are formed a second time afterwards
        // -- two multiplications against the 460 of the inversion -- and the Z's are fetched again.
        uint32_t a[16], b[16], inv[16], z[16];
        auto products = [&]() {
            get7(0 * 21 + 14, a);
            get7(1 * 21 + 14, z);
            F::mul_k(a, z, a);
            get7(2 * 21 + 14, z);
            F::mul_k(a, z, b);
            get7(3 * 21 + 14, z);
        };
        products();
        F::mul_k(b, z, inv);
        F::invert(inv, inv);
        products();
        uint32_t i1[16], i2[16], i3[16], i4[16];
        F::mul_k(inv, b, i4);                       // 1 / Z4
        F::mul_k(inv, z, inv);                      // 1 / (Z1 Z2 Z3)      (z = Z4)
        F::mul_k(inv, a, i3);
        get7(2 * 21 + 14, z);
        F::mul_k(inv, z, inv);                      // 1 / (Z1 Z2)
        get7(0 * 21 + 14, z);
        F::mul_k(inv, z, i2);
        get7(1 * 21 + 14, z);
        F::mul_k(inv, z, i1);
        finish(0, i1);
        finish(1, i2);
        finish(2, i3);
        finish(3, i4);
    }
};

// Affine, canonical coordinates of R (ecnXXXget: edwards.c:221-239).  X and Y wait in the first table slot -- the table is dead
// by now -- while Z is inverted, so that the inversion has the register file to itself.
template <class TAB>
MA_DEV void ed448_affine_words(Ed28::Ext& R, const TAB& T, uint64_t* xw, uint64_t* yw) {
    using F = Fe28;
    Ed448Slots<TAB> S{T};
    S.put7(0, R.X);
    S.put7(7, R.Y);
    uint32_t zi[16], c[16];
    F::invert(R.Z, zi);
    S.get7(0, c);
    F::mul_k(c, zi, c);
    F::to_words(c, xw);
    S.get7(7, c);
    F::mul_k(c, zi, c);
    F::to_words(c, yw);
}

// One fused ED448 scalar multiplication + affine export.  ew: the scalar as seven little-endian words; X, Y, Z: 8 x 56-bit
// limbs each; tab: this lane's table slots, word k at tab[k * tstride]; xw, yw: canonical affine coordinates, seven words.
template <bool FINAL_T = false, class TAB, class DIG>         // FINAL_T: the sum leaves with its T coordinate (a further addition follows)
MA_DEV void ed448_mul_acc(DIG& dig, const spint* X, const spint* Y, const spint* Z, const TAB& T, Ed28::Ext& R) {
    using E = Ed28;
    using F = Fe28;

    {   // ---- table: projective P -> extended; 2P, 3P, 4P; one shared inversion; cached affine form.
        // Round 4: ONE point is live at any time.  Each multiple is STASHED in its own table slot as canonical
        // (X, Y, Z) -- 3 x 7 words, exactly the size of an entry -- as soon as it exists; the products of the Z's are formed from the
        // slots afterwards (a = Z1 Z2, b = a Z3, c = b Z4), and after the one inversion every slot is fetched, scaled and overwritten with its
        // affine entry (x, y, 39081 x y).  (Round 3 kept Q, 2P, 3P, 4P -- 256 VGPRs -- to the end: 600 spilled registers.)
        Ed448Slots<TAB> S{T};
        {
            E::Ext Q;
            {
                uint32_t px[16], py[16], pz[16];
                E::from56(X, px);
                E::from56(Y, py);
                E::from56(Z, pz);
                F::mul_k(px, pz, Q.X);              // (XZ : YZ : Z^2 : XY)
                F::mul_k(py, pz, Q.Y);
                F::sqr_k(pz, Q.Z);
                F::mul_k(px, py, Q.T);
            }
            S.stash(Q, 0);
            S.put7(3 * 21, Q.T);                    // (the fourth slot is free until 4P exists)
            E::dbl<true>(Q);                        // 2P, with T (the addition below reads it)
            S.stash(Q, 1);
            E::add_ext_fetched(Q, [&](int c, uint32_t* out) { S.get7(c == 3 ? 3 * 21 : c * 7, out); });     // 3P = 2P + P
            S.stash(Q, 2);
            S.get7(1 * 21, Q.X);                      // 2P again (the doubling reads X, Y, Z only)
            S.get7(1 * 21 + 7, Q.Y);
            S.get7(1 * 21 + 14, Q.Z);
            E::dbl<false>(Q);                       // 4P
            S.stash(Q, 3);
        }
        S.to_affine_entries();
    }

    F::set(0, R.X);
    F::set(1, R.Y);
    F::set(1, R.Z);
    F::set(0, R.T);

#pragma unroll 1
    for (int i = 0; i < 150; i++) {
        const uint32_t win = dig.window(i);                 // e' = e + sum 4*8^i: window - 4 is the signed digit
        const int dgt = (int)win - 4;                       // [-4, 3]
        const bool neg = dgt < 0;
        const uint32_t m = (uint32_t)(neg ? -dgt : dgt);    // 0..4
        if (i != 0) {
#pragma unroll 1
            for (int j = 0; j < 3; j++) E::dbl(R, j == 2);
        }
        // constant-time lookup: every entry is read; start from the neutral element (x, y, td) = (0, 1, 0)
        uint64_t sel[21];
        static_for<0, 21>([&](auto K) { sel[K] = (K == 7) ? 1u : 0u; });
        // (all 84 loads are issued before the first select: one memory latency per window, and the doublings' temporaries
        // are dead here, so the 168 landing registers are free)
        // The memory clobber keeps the (loop-invariant) loads inside the iteration: hoisted out of the loop they would
        // occupy 168 registers across the doublings, i.e. be spilled and reloaded from scratch one by one, each with
        // its own s_waitcnt vmcnt(0) -- measured as 40 % of the kernel's time in SQ_WAIT_ANY.
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" ::: "memory");
#endif
        // two entries (84 landing registers) per round: four at once do not fit next to the point
#pragma unroll 1
        for (int e = 0; e < 4; e += 2) {
            uint64_t ent[2][21];
            const uint64_t* tab = T.origin() + (size_t)(e * 21) * T.stride();
            const size_t tstride = T.stride();
            static_for<0, 2>([&](auto EI) {
                static_for<0, 21>([&](auto K) { ent[EI][K] = tab[(size_t)(EI * 21 + K) * tstride]; });
            });
            static_for<0, 2>([&](auto EI) {
                const bool hit = (m == (uint32_t)(e + EI + 1));
                static_for<0, 21>([&](auto K) {
                    const uint64_t a = ent[EI][K], b = sel[K];
                    sel[K] = hit ? a : b;
                });
            });
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" ::: "memory");
#endif
        }
        // -Q = (-x, y, -td)
        uint32_t xs[16], ys[16], ts[16], nx[16], nt[16];
        F::from_words(sel, xs);
        F::from_words(sel + 7, ys);
        F::from_words(sel + 14, ts);
        E::neg2p(xs, nx);
        E::neg2p(ts, nt);
        F::select(neg, xs, nx, xs);
        F::select(neg, ts, nt, ts);
        E::add_cached(R, xs, ys, ts, FINAL_T && i == 149);
    }
}
template <class TAB, class DIG>
MA_DEV void ed448_mul_get_one(DIG& dig, const spint* X, const spint* Y, const spint* Z, const TAB& T, uint64_t* xw, uint64_t* yw) {
    Ed28::Ext R;
    ed448_mul_acc<false>(dig, X, Y, Z, T, R);
    // ---- affine, canonical (ecnXXXget: edwards.c:221-239)
    ed448_affine_words(R, T, xw, yw);
}
// (scalar words, table as a strided array: the form tools/fe_host_check.hip runs on the CPU)
MA_DEV void ed448_mul_get_one(const uint64_t* ew, const spint* X, const spint* Y, const spint* Z, uint64_t* tab, size_t tstride,
                              uint64_t* xw, uint64_t* yw) {
    Win3Regs dig;
    dig.init(ew);
    ed448_mul_get_one(dig, X, Y, Z, TabStrided{tab, tstride}, xw, yw);
}

// Fused double multiplication + affine export for ED448: the affine coordinates of e*P + f*Q (ecnXXXmul2 followed by
// ecnXXXget, the verification pattern ed448.c:305).  Both scalars in 225 signed 2-bit digits (e' = e + sum 2*4^i,
// digit = window - 2 in [-2, 1]); the tables {P, 2P}, {Q, 2Q} take the four entry slots of the same per-lane workspace
// as ed448_mul_get_one; per window two doublings and two additions (one rolled copy of each in the instruction stream).
template <class TAB, class DIG>
MA_DEV void ed448_mul2_get_one(DIG& dige, const spint* PX, const spint* PY, const spint* PZ,
                               DIG& digf, const spint* QX, const spint* QY, const spint* QZ,
                               const TAB& T, uint64_t* xw, uint64_t* yw) {
    using E = Ed28;
    using F = Fe28;
    E::Ext R;
    {
        // (round 4: one point live at a time; see Ed448Slots)
        Ed448Slots<TAB> S{T};
        auto ext2 = [&](const spint* X, const spint* Y, const spint* Z, int entry) {
            E::Ext A;
            {
                uint32_t px[16], py[16], pz[16];
                E::from56(X, px);
                E::from56(Y, py);
                E::from56(Z, pz);
                F::mul_k(px, pz, A.X);
                F::mul_k(py, pz, A.Y);
                F::sqr_k(pz, A.Z);
            }
            S.stash(A, entry);
            E::dbl<false>(A);
            S.stash(A, entry + 1);
        };
        ext2(PX, PY, PZ, 0);
        ext2(QX, QY, QZ, 2);
        S.to_affine_entries();
    }
    F::set(0, R.X);
    F::set(1, R.Y);
    F::set(1, R.Z);
    F::set(0, R.T);
#pragma unroll 1
    for (int i = 0; i < 225; i++) {
        if (i != 0) {
#pragma unroll 1
            for (int j = 0; j < 2; j++) E::dbl(R, j == 1);
        }
        const int de = (int)dige.window(i) - 2, df = (int)digf.window(i) - 2;       // [-2, 1]
#pragma unroll 1
        for (int which = 0; which < 2; which++) {               // 0: digit of e, table {P, 2P};  1: digit of f, table {Q, 2Q}
            const int dgt = which ? df : de;
            const bool neg = dgt < 0;
            const uint32_t m = (uint32_t)(neg ? -dgt : dgt);
            uint64_t sel[21];
            static_for<0, 21>([&](auto K) { sel[K] = (K == 7) ? 1u : 0u; });
            uint64_t ent[2][21];
            const uint64_t* tab = T.origin() + (size_t)(2 * which * 21) * T.stride();
            const size_t tstride = T.stride();
            static_for<0, 2>([&](auto EI) {
                static_for<0, 21>([&](auto K) { ent[EI][K] = tab[(size_t)(EI * 21 + K) * tstride]; });
            });
            static_for<0, 2>([&](auto EI) {
                const bool hit = (m == (uint32_t)(EI + 1));
                static_for<0, 21>([&](auto K) {
                    const uint64_t a = ent[EI][K], b = sel[K];
                    sel[K] = hit ? a : b;
                });
            });
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" ::: "memory");
#endif
            uint32_t xs[16], ys[16], ts[16], nx[16], nt[16];
            F::from_words(sel, xs);
            F::from_words(sel + 7, ys);
            F::from_words(sel + 14, ts);
            E::neg2p(xs, nx);
            E::neg2p(ts, nt);
            F::select(neg, xs, nx, xs);
            F::select(neg, ts, nt, ts);
            E::add_cached(R, xs, ys, ts, which == 0);
        }
    }
    ed448_affine_words(R, T, xw, yw);
}

MA_DEV void ed448_mul2_get_one(const uint64_t* ew, const spint* PX, const spint* PY, const spint* PZ,
                               const uint64_t* fw, const spint* QX, const spint* QY, const spint* QZ,
                               uint64_t* tab, size_t tstride, uint64_t* xw, uint64_t* yw) {
    Win2Regs de, df;
    de.init(ew);
    df.init(fw);
    ed448_mul2_get_one(de, PX, PY, PZ, df, QX, QY, QZ, TabStrided{tab, tstride}, xw, yw);
}

// Fused GENERATOR multiplication + affine export for ED448 (see ed26.h ed25519_mulgen_get_one): ED448_KEY_PAIR and
// ED448_SIGN open with ecnXXXgen, ecnXXXmul, ecnXXXget (ed448.c:167-184, 196-199).  W = 4: e' = e + sum_{i<113} 8*16^i, 113
// signed 4-bit digits, the 113 x 8 multiples m * 16^i * G precomputed as (x, y, 39081 x y) in sixteen 28-bit limbs
// (generated/comb_ED448.h, 173 568 bytes, wave-uniform reads; W = 5 measured the same), one complete mixed addition per window, no doublings.
template <class TAB, bool INIT = true>         // INIT = false: R += e*G (R holds a sum with its T coordinate)
MA_DEV void ed448_mulgen_acc(const uint64_t* ew, Ed28::Ext& R) {
    using E = Ed28;
    using F = Fe28;
    constexpr int W = TAB::W, NW = TAB::NW, E2 = 1 << (W - 1);       // window width, windows, entries per window
    static_assert(W * NW >= 449 && W * NW <= 512, "e + bias must fit the windows and eight words");
    uint64_t w[8];
    {
        constexpr auto cw = [](int k) {
            uint64_t v = 0;
            for (int b = 0; b < 64; b++) {
                const int pos = 64 * k + b;
                if (pos < W * NW && pos % W == W - 1) v |= (uint64_t)1 << b;
            }
            return v;
        };
        unsigned __int128 acc = 0;
        static_for<0, 8>([&](auto K) {
            constexpr int k = K;
            acc += (unsigned __int128)(k < 7 ? ew[k < 7 ? k : 0] : 0) + cw(k);
            w[k] = (uint64_t)acc;
            acc >>= 64;
        });
    }
    if constexpr (INIT) {
        F::set(0, R.X);
        F::set(1, R.Y);
        F::set(1, R.Z);
        F::set(0, R.T);
    }
#pragma unroll 1
    for (int i = 0; i < NW; i++) {
        const int dgt = (int)((uint32_t)w[0] & (uint32_t)(2 * E2 - 1)) - E2;        // [-2^(W-1), 2^(W-1) - 1]
        static_for<0, 8>([&](auto K) {
            constexpr int k = K;
            w[k] >>= W;
            if constexpr (k < 7) w[k] |= w[k + 1] << (64 - W);
        });
        const bool neg = dgt < 0;
        const uint32_t m = (uint32_t)(neg ? -dgt : dgt);        // 0 .. 2^(W-1)
        uint32_t sel[3][16];
        static_for<0, 3>([&](auto CI) { static_for<0, 16>([&](auto K) { sel[CI][K] = (CI == 1 && K == 0) ? 1u : 0u; }); });
        // (selection as OR of masked entries, see wn26.h; limb 0 of y starts at 1 only for a zero digit)
        sel[1][0] = (m == 0) ? 1u : 0u;
        static_for<0, E2>([&](auto MM) {
            constexpr int mm = MM;
            uint32_t mask = (m == (uint32_t)(mm + 1)) ? 0xffffffffu : 0u;
#if defined(__HIP_DEVICE_COMPILE__)
            asm("" : "+v"(mask));       // opaque: otherwise the compiler turns (entry & mask) back into a select with a move
#endif
            static_for<0, 3>([&](auto CI) {
                static_for<0, 16>([&](auto K) { sel[CI][K] |= (uint32_t)TAB::get(((i * E2 + mm) * 3 + CI) * 16 + K) & mask; });
            });
        });
        uint32_t nx[16], nt[16];
        E::neg2p(sel[0], nx);
        E::neg2p(sel[2], nt);
        F::select(neg, sel[0], nx, sel[0]);
        F::select(neg, sel[2], nt, sel[2]);
        E::add_cached(R, sel[0], sel[1], sel[2], true);
    }
}
template <class TAB>
MA_DEV void ed448_mulgen_get_one(const uint64_t* ew, uint64_t* xw, uint64_t* yw) {
    using F = Fe28;
    Ed28::Ext R;
    ed448_mulgen_acc<TAB>(ew, R);
    uint32_t zi[16], ax[16], ay[16];
    F::invert(R.Z, zi);
    F::mul_k(R.X, zi, ax);
    F::mul_k(R.Y, zi, ay);
    F::to_words(ax, xw);
    F::to_words(ay, yw);
}
// Fused e*G + f*Q + affine export for ED448: ED448_VERIFY's ecnXXXmul2(&G, &Q, ...) + ecnXXXget (ed448.c:290-310; the first
// point is the generator).  f*Q as in ed448_mul_get_one (the last addition also produces T), then e*G through the fixed-base
// table (ed448_mulgen_acc), see ed26.h ed25519_mulgen2_get_one.
template <class COMB, class TAB, class DIG>
MA_DEV void ed448_mulgen2_get_one(const uint64_t* ew, DIG& digf, const spint* QX, const spint* QY, const spint* QZ,
                                  const TAB& T, uint64_t* xw, uint64_t* yw) {
    Ed28::Ext R;
    ed448_mul_acc<true>(digf, QX, QY, QZ, T, R);
    ed448_mulgen_acc<COMB, false>(ew, R);
    ed448_affine_words(R, T, xw, yw);
}
template <class COMB>
MA_DEV void ed448_mulgen2_get_one(const uint64_t* ew, const uint64_t* fw, const spint* QX, const spint* QY, const spint* QZ,
                                  uint64_t* tab, size_t tstride, uint64_t* xw, uint64_t* yw) {
    Win3Regs df;
    df.init(fw);
    ed448_mulgen2_get_one<COMB>(ew, df, QX, QY, QZ, TabStrided{tab, tstride}, xw, yw);
}

// rfc7748() on the BASE POINT u = 5 of X448 (public-key generation, rfc7748.c:297-333).  ED448 (x^2 + y^2 = 1 - 39081 x^2 y^2) is
// 4-isogenous to curve448 with (u, v) = (y^2 / x^2, ...) (RFC 7748 section 4.2), and its generator maps to u = 5: the isogeny is
// a group homomorphism, so [k](5) = Y^2 / X^2 of k*G, k*G from the fixed-base table (ed448_mulgen_acc).  k clamped as
// rfc7748.c:135-141; k = 4q gives the neutral element, X = 0, and 0^(p-2) = 0 makes the result 0 as the ladder's.
template <class TAB>
MA_DEV void x448_base_one(const uint64_t* kw_in, uint64_t* ow) {
    using F = Fe28;
    uint64_t kw[7];
    static_for<0, 7>([&](auto K) { kw[K] = kw_in[K]; });
    kw[0] &= ~3ull;
    kw[6] |= 0x8000000000000000ull;
    Ed28::Ext R;
    ed448_mulgen_acc<TAB>(kw, R);
    uint32_t x2[16], y2[16], xi[16], u[16];
    F::sqr_k(R.X, x2);
    F::sqr_k(R.Y, y2);
    F::invert(x2, xi);
    F::mul_k(y2, xi, u);
    F::to_words(u, ow);
}

// (Rounds 2-4 ran TWO scalars per lane with one inversion -- ed448_mulgen_get_two / x448_base_two, the first result parked in LDS -- and
// spilled 14 / 2 registers in that epilogue; round 5 shares the inversion between up to 32 records instead: csrc/edlad_k.h,
// capi_ED448G.hip k_ed448_mulgen / k_x448_base.)

}  // namespace ma
