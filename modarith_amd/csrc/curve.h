// modarith_amd/csrc/curve.h -- the part of the reference's curve layer that is the same for edwards.c and
// weierstrass.c (cmv, cmp, ran, sub, select, the signed 4-bit fixed-window mul, mul2), written once over a
// curve class K that supplies the formulas (add, dbl, neg, inf, isinf, affine, setxy, gen, cof), plus the
// batched kernels of the curve API (curve.h:13-29).  One point per lane, coordinates in VGPRs, built from the
// bit-exact Field<P> functions in the reference's order.  The 9-entry window table of ecnXXXmul does not fit the
// register file (9 x 3 x N limbs); it lives in the wave's slab of a global workspace laid out
// [wave][entry][coord][limb][lane] (coalesced 512-byte rows; sized to the resident grid so it stays in the 256 MiB
// Infinity Cache) and is scanned in full on every lookup: the table index never forms an address, selection is
// lane-predicated modcmv.  The recoded scalar digits live in LDS (one byte per window per lane).
#pragma once
#include "field.h"
#include "kernels.h"

namespace ma {

// scalar(i): i again, pinned into an SGPR.  The window counters of the multiplication loops are wave-uniform, but feed the LDS address of
// the window's digit, so the compiler may keep them in a VGPR and close the loop with a carry-out vote (v_subrev_co + s_cbranch_vccz):
// harmless, yet indistinguishable in the ISA from a branch on lane data (tools/ct_audit.py).  Pinned, the loop closes on s_cmp / SCC.
MA_DEV int scalar(int i) {
#if defined(__HIP_DEVICE_COMPILE__)          // (the host pass of tools/fe_host_check.hip sees this header too)
    asm volatile("" : "+s"(i));
#endif
    return i;
}
// ---- the limb contract of the scalar-multiplication kernels (round 6).  The reference's ecnXXXmul over the pasted field.c returns
// DEFINED limbs for every 64-bit limb pattern (edwards.c:435-482: its products wrap at 128 bits, nothing else happens).  The classes
// the kernels are built from are narrower: the FAST products of Field<P, true> and the resident half-limb forms (fh51.h: a limb of
// 2^58 loses its top bits at load; fh56.h) equal the reference for limbs inside the budget 2^(Radix+2) that every field function's
// output keeps (kernels.h in_limb_budget) and not beyond.  The kernels therefore take the vote OpMulAuto takes (kernels.h:117-124) at
// point load, per wave and pass: a wave with a limb beyond the budget leaves its points untouched; a second launch, built from the EXACT
// class of the same curve (Field<P, false>: the reference's rows for all inputs), takes the same vote and computes exactly those passes.  No legitimate point -- nothing a field or curve function returns -- ever takes the second path.
template <class F> struct field_is_exact : std::false_type {};                         // resident forms: never
template <class P, bool PIN> struct field_is_exact<Field<P, false, PIN>> : std::true_type {};
template <class P, bool PIN> struct field_is_exact<Field<P, true, PIN>> : std::integral_constant<bool, !Field<P, true, PIN>::FAST && !Field<P, true, PIN>::SPLIT4> {};
template <class Crv> struct exact_class;                                               // the same curve on Field<P, false>: edwards.h, weierstrass.h

template <class Crv, class P_, class F_ = Field<P_, true>>
struct CurveOps {
    using P = P_;
    using F = F_;               // Field<P, true>: FAST product path where the driver proved it (P::SPLIT > 0), else exact; or a resident form (fh51.h)
    using limb_t = typename F::limb_t;
    static constexpr int N = P::N;             // limbs of a coordinate in HBM and in the window tables
    static constexpr int NL = F::NL;           // words of a coordinate in registers
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
    struct Point { limb_t x[NL], y[NL], z[NL]; };

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
        limb_t a[NL], b[NL];
        F::modmul(p.x, q.z, a);
        F::modmul(q.x, p.z, b);
        int eq = F::modcmp(a, b);
        F::modmul(p.y, q.z, a);
        F::modmul(q.y, p.z, b);
        return eq & F::modcmp(a, b);
    }

    // ---- window tables in the global workspace.  Every wave (= workgroup of the scalar-multiplication kernels) owns one slab
    // [entry][coord][limb][64 lanes]: the slab base is wave-uniform (SGPRs), a lane adds 8 * lane, and entry / coordinate / limb
    // are compile-time or loop-uniform offsets -- the loads of a table scan need no per-lane address arithmetic (one
    // global_load_dwordx2 v, v_lane, s[base] offset:imm per limb), and each is one contiguous 512-byte row.
    static constexpr size_t ENTRY_WORDS = (size_t)3 * N * 64;            // one table entry of one wave
    static constexpr size_t TABLE_WORDS = 2 * 9 * 3 * N;                 // per lane: room for the two tables of mul2
    static constexpr size_t SLAB_WORDS = 64 * TABLE_WORDS;               // per wave
    struct Table {
        spint* base;       // slab of this wave (table of mul, or first / second table of mul2): wave-uniform
        unsigned lane;     // 0..63
        // here(): the lane offset as a value born at this access -- otherwise the row addresses (an entry of an 8- or 9-limb curve
        // spans 12-14 KB, beyond one base register's +-4 KB immediate range) are loop-invariant, get hoisted to the top of the
        // kernel and live (and spill) through the whole multiplication; recomputed they cost two instructions per access
        MA_DEV unsigned here() const {
            unsigned l = lane;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(l));
#endif
            return l;
        }
        MA_DEV void put(int k, const Point& w) const {
            spint* q = base + (size_t)k * ENTRY_WORDS + here();
            static_for<0, N>([&](auto I) {
                q[(0 * N + I) * 64] = F::pack(w.x, I);
                q[(1 * N + I) * 64] = F::pack(w.y, I);
                q[(2 * N + I) * 64] = F::pack(w.z, I);
            });
        }
        MA_DEV void get(int k, Point& w) const {
            const spint* q = base + (size_t)k * ENTRY_WORDS + here();
            static_for<0, N>([&](auto I) {
                F::unpack(q[(0 * N + I) * 64], w.x, I);
                F::unpack(q[(1 * N + I) * 64], w.y, I);
                F::unpack(q[(2 * N + I) * 64], w.z, I);
            });
        }
        MA_DEV Table second() const { return Table{base + 9 * ENTRY_WORDS, lane}; }
    };

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
        // p is read back from W[1] where an odd entry needs it: no point stays live across a formula it is not part of
#pragma unroll 1
        for (int k = 2; k <= 8; k++) {
            if (k & 1) { Point P1; W.get(k - 1, T); W.get(1, P1); Crv::add(P1, T); }
            else       { W.get(k >> 1, T); Crv::dbl(T); }
            W.put(k, T);
        }
    }

    // Signed 4-bit recoding of a scalar (edwards.c:452-467): w_j = nibble_j + c_j - 16 c_{j+1}, c_0 = 0, c_{j+1} = (nibble_j + c_j > 7),
    // j = 0 .. 2*NB-1, and w_{2*NB} = the final carry.  The digits are written once, before the point is touched, into the lane's
    // column of an LDS array dg[j * 64] (one byte each, 2*NB+1 of them: 4-8.5 KB per wave) and read back one per window with a
    // ds_read_i8 -- no scalar words or carry masks stay in registers during the multiplication.  Each lane reads only what it wrote
    // itself (the kernels run one wave per workgroup), so no barrier is involved.
    static constexpr int NDIG = 2 * NB + 1;
    static MA_DEV void recode(const spint* ew, signed char* dg) {
        unsigned c = 0;
        static_for<0, NW>([&](auto K) {
            constexpr int cnt = (2 * NB - 16 * K) < 16 ? (2 * NB - 16 * K) : 16;
            spint word = ew[K];
            signed char* d = dg + (size_t)(16 * K) * 64;
#pragma unroll 1
            for (int j = 0; j < cnt; j++) {
                const unsigned v = (unsigned)(word & 15) + c;
                c = v > 7 ? 1u : 0u;
                d[j * 64] = (signed char)((int)v - (int)(c << 4));
                word >>= 4;
            }
        });
        dg[(size_t)(2 * NB) * 64] = (signed char)c;
    }

    // P = e*P, signed 4-bit fixed window (edwards.c:435-482).  dg = the recoded scalar (recode() above).  The table lookup of a
    // window comes after its four doublings (the order does not matter to either), so the looked-up point is not live across them.
    static MA_DEV void mul(const signed char* dg, Point& p, const Table& W) {
        build_table(p, W);
        select((int)dg[(size_t)(2 * NB) * 64], W, p);
#pragma unroll 1
        for (int i = 2 * NB - 1; i >= 0; i--) {
            i = scalar(i);
#pragma unroll 1
            for (int r = 0; r < 4; r++) Crv::dbl(p);
            Point Q;
            select((int)dg[(size_t)i * 64], W, Q);
            Crv::add(Q, p);
        }
    }

    // R = e*P + f*Q (edwards.c:486-510).  The reference walks a joint sparse form (dnaf, 404-431) with data-dependent
    // branches ("not constant time"), which would diverge across lanes.  Here: two signed 4-bit fixed-window
    // multiplications sharing their doublings -- tables {0..8}P and {0..8}Q in the workspace, per window four
    // doublings and two complete additions -- so every lane runs the same 2*NB windows and the multiplication is
    // constant-time as well.  Same point as the reference's, another projective representative.
    // de / df = the recoded scalars (two digit columns in LDS).
    static MA_DEV void mul2(const signed char* de, const signed char* df, Point& r, const Table& W) {
        // the tables {0..8}P (W) and {0..8}Q (W.second()) are built by the caller, one point loaded at a time (k_ed_mul2); the two
        // halves (lookup, addition) run through a loop of two rolled iterations, so the instruction stream holds one copy of
        // add / dbl / select, as in mul; t is wave-uniform
        Crv::inf(r);
#pragma unroll 1
        for (int i = 2 * NB; i >= 0; i--) {
            i = scalar(i);
            if (i != 2 * NB) {
#pragma unroll 1
                for (int k = 0; k < 4; k++) Crv::dbl(r);
            }
#pragma unroll 1
            for (int t = 0; t < 2; t++) {
                Point T;
                select((int)(t ? df : de)[(size_t)i * 64], t ? W.second() : W, T);
                Crv::add(T, r);                  // the first pass adds to the neutral element
            }
        }
    }

    // R = e*P + f*Q with the REFERENCE'S OWN walk (edwards.c:404-431 dnaf, 486-510; weierstrass.c:545-569): the joint sparse form
    // w[k] = (bit_k(3e) - bit_k(e)) + 3 (bit_k(3f) - bit_k(f)) in {-4..4}, table W = {O, P, Q-P, Q, Q+P}, R = O, and from the first
    // non-zero digit down to k = 1: R = 2R, then R += W[w] or R -= W[-w].  Same field calls in the same order, hence the reference's
    // projective limbs -- and, like the reference ("not constant time"), a walk that depends on the scalars: lanes of a wave start
    // at different digits and skip different additions, so the wave pays for the union of their paths (measured in round 3: 4-18 % slower than mul2 above).
    // jsf_digits() produces the digits from the top by shifting e, 3e, f, 3f left one bit per step and packs them, biased by 4, two
    // per byte into the lane's LDS column (4*NB+4 bytes) BEFORE any point is loaded: the four multi-word shift registers (up to 72
    // VGPRs for the 521-bit field) are dead by the time the walk starts.
    static constexpr int JSF_TOP = 8 * NB + 7;                    // index of the highest digit (edwards.c:497)
    static constexpr int JSF_BYTES = (JSF_TOP + 1) / 2;
    static MA_DEV void jsf_digits(const spint* ew, const spint* fw, unsigned char* dj) {
        constexpr int NX = NW + 1;                        // 3e needs two more bits than e
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
        constexpr int S = NX * 64 - 1 - JSF_TOP;          // left-align: digit TOP in bit 63 of the top word
        shl_words<S, NX>(a); shl_words<S, NX>(a3); shl_words<S, NX>(b); shl_words<S, NX>(b3);
#pragma unroll 1
        for (int q = 0; q < JSF_BYTES; q++) {             // digits TOP-2q and TOP-2q-1 (the last byte's low half is digit 0: unused)
            unsigned byte = 0;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int j = ((int)(a3[NX - 1] >> 63) - (int)(a[NX - 1] >> 63)) + 3 * ((int)(b3[NX - 1] >> 63) - (int)(b[NX - 1] >> 63));
                shl_words<1, NX>(a); shl_words<1, NX>(a3); shl_words<1, NX>(b); shl_words<1, NX>(b3);
                byte = (byte << 4) | (unsigned)(j + 4);
            }
            dj[(size_t)q * 64] = (unsigned char)byte;
        }
    }
    static MA_DEV void mul2_exact(const unsigned char* dj, const Point& p, const Point& q, Point& r, const Table& W) {
        Point T;
        Crv::inf(T); W.put(0, T);
        W.put(1, p);
        W.put(3, q);
        // W[2] = Q - P (ecnXXXsub = copy, negate, add: edwards.c:114-119), W[4] = Q + P: one rolled loop, one copy of add
#pragma unroll 1
        for (int t = 0; t < 2; t++) {
            Point w;
            W.get(1, w);
            if (t == 0) Crv::neg(w);
            W.get(3, T);
            Crv::add(w, T);
            W.put(2 + 2 * t, T);
        }
        Crv::inf(r);
        bool started = false;
#pragma unroll 1
        for (int i = JSF_TOP; i >= 1; i--) {
            const int pos = JSF_TOP - i;
            const unsigned byte = dj[(size_t)(pos >> 1) * 64];
            const int j = (int)((pos & 1) ? (byte & 15u) : (byte >> 4)) - 4;
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
        spint x[N], y[N], z[N];
        static_for<0, N>([&](auto I) {
            x[I] = Pb[((size_t)(0 * N + I)) * ld + j];
            y[I] = Pb[((size_t)(1 * N + I)) * ld + j];
            z[I] = Pb[((size_t)(2 * N + I)) * ld + j];
        });
        F::from_limbs(x, p.x); F::from_limbs(y, p.y); F::from_limbs(z, p.z);
    }
    // every limb of point j inside the budget the field functions of this class are exact for?  (3 N loads that the load proper hits in L2)
    static MA_DEV bool limbs_ok(const spint* Pb, size_t ld, size_t j) {
        bool ok = true;
        static_for<0, 3>([&](auto C) {
            spint v[N];
            static_for<0, N>([&](auto I) { v[I] = Pb[((size_t)(C * N + I)) * ld + j]; });
            ok = ok & in_limb_budget<P>(v);                 // (&, not &&: one vote per pass is the only branch this makes)
        });
        return ok;
    }
    static MA_DEV void store(spint* Pb, size_t ld, size_t j, const Point& p) {
        spint x[N], y[N], z[N];
        F::to_limbs(p.x, x); F::to_limbs(p.y, y); F::to_limbs(p.z, z);
        static_for<0, N>([&](auto I) {
            Pb[((size_t)(0 * N + I)) * ld + j] = x[I];
            Pb[((size_t)(1 * N + I)) * ld + j] = y[I];
            Pb[((size_t)(2 * N + I)) * ld + j] = z[I];
        });
    }
};

// ---------------------------------------------------------------- kernels
// resident waves per SIMD the scalar-multiplication kernels are register-budgeted for (256 VGPRs each); three waves at
// 170 VGPRs measured the same throughput, one wave at 512 VGPRs 20 % less
#ifndef MA_MUL_WPS
#define MA_MUL_WPS 2
#endif
// GUARD (the limb contract, see the top of this file): 0 = the class is exact for every limb pattern, no vote; +1 = the fast class: a
// pass in which some lane's point has a limb beyond the budget is skipped, its points left as they are; -1 = the exact class, launched
// behind it on the same batch: it takes the same vote and computes exactly the passes that have such a point.  It sees the fast
// kernel's OUTPUTS where ecn mul worked in place, and votes them inside the budget -- as every output of a field function is (the
// invariant the whole library rests on; modlimbs / Curve.limbs_ok test it) -- so no pass is computed twice and none is left out.
// The vote is on the POINT's limbs; no scalar bit ever reaches a branch (tools/ct_allowlist.json).
#if defined(__HIP_DEVICE_COMPILE__)
#define MA_WAVE_ALL(x) __all(x)
#else
#define MA_WAVE_ALL(x) (x)
#endif
// One wave per workgroup; the workgroup's LDS holds the recoded scalars of its 64 lanes (CurveOps::recode), the workspace slab
// blockIdx.x its window tables.
template <class Crv, int GUARD = 0>
__global__ __launch_bounds__(64, (GUARD < 0 ? 2 : MA_MUL_WPS)) void k_ed_mul(const unsigned char* e, spint* Pb, size_t n, size_t ld, spint* ws) {
    using E = Crv;
    // the LDS of a CU (160 KB on gfx950) must hold the digit arrays of all its resident workgroups (4 SIMDs x MA_MUL_WPS waves), for the
    // double multiplications too (two / JSF arrays per workgroup): ADVICE of round 4
    static_assert((size_t)4 * MA_MUL_WPS * 2 * E::NDIG * 64 <= (size_t)160 * 1024, "LDS budget of the resident scalar-multiplication grid");
    __shared__ signed char digs[E::NDIG * 64];
    const typename E::Table W{ws + (size_t)blockIdx.x * E::SLAB_WORDS, threadIdx.x};
    signed char* dg = digs + threadIdx.x;
    // base: the wave's first element of this pass, wave-uniform (SGPRs); a lane's element index base + lane is formed where it is used
    // (W.here(): the lane number as a fresh value), so no 64-bit index or address stays in VGPRs across the multiplication
    for (size_t base = (size_t)blockIdx.x * 64; base < n; base += (size_t)gridDim.x * 64) {
        if (base + W.here() >= n) continue;
        if constexpr (GUARD != 0) {
            if ((MA_WAVE_ALL(E::limbs_ok(Pb, ld, base + W.here())) != 0) != (GUARD > 0)) continue;
        }
        {
            spint ew[E::NW];
            load_be_record<typename E::P>(e, base + W.here(), ew);         // big-endian byte record -> little-endian words
            E::recode(ew, dg);
        }
        typename E::Point p;
        E::load(Pb, ld, base + W.here(), p);
        E::mul(dg, p, W);
        E::store(Pb, ld, base + W.here(), p);
    }
}

template <class Crv, int GUARD = 0>
__global__ __launch_bounds__(64, (GUARD < 0 ? 2 : MA_MUL_WPS)) void k_ed_mul2(const unsigned char* e, const spint* Pb, const unsigned char* f, const spint* Qb, spint* Rb,
                                                size_t n, size_t ld, spint* ws) {
    using E = Crv;
    __shared__ signed char digs[2 * E::NDIG * 64];
    const typename E::Table W{ws + (size_t)blockIdx.x * E::SLAB_WORDS, threadIdx.x};
    signed char* de = digs + threadIdx.x;
    signed char* df = de + E::NDIG * 64;
    for (size_t base = (size_t)blockIdx.x * 64; base < n; base += (size_t)gridDim.x * 64) {
        if (base + W.here() >= n) continue;
        if constexpr (GUARD != 0) {
            const bool ok = ((int)E::limbs_ok(Pb, ld, base + W.here()) & (int)E::limbs_ok(Qb, ld, base + W.here())) != 0;
            if ((MA_WAVE_ALL(ok) != 0) != (GUARD > 0)) continue;
        }
        {
            spint ew[E::NW];
            load_be_record<typename E::P>(e, base + W.here(), ew);
            E::recode(ew, de);
            load_be_record<typename E::P>(f, base + W.here(), ew);
            E::recode(ew, df);
        }
#pragma unroll 1
        for (int h = 0; h < 2; h++) {                        // one point at a time: load, build its table, forget
            typename E::Point p;
            E::load(h ? Qb : Pb, ld, base + W.here(), p);
            E::build_table(p, h ? W.second() : W);
        }
        typename E::Point r;
        E::mul2(de, df, r, W);
        E::store(Rb, ld, base + W.here(), r);
    }
}

template <class Crv, int GUARD = 0>
__global__ __launch_bounds__(64, (GUARD < 0 ? 2 : MA_MUL_WPS)) void k_ed_mul2x(const unsigned char* e, const spint* Pb, const unsigned char* f, const spint* Qb, spint* Rb,
                                                 size_t n, size_t ld, spint* ws) {
    using E = Crv;
    __shared__ unsigned char digs[E::JSF_BYTES * 64];
    const typename E::Table W{ws + (size_t)blockIdx.x * E::SLAB_WORDS, threadIdx.x};
    unsigned char* dj = digs + threadIdx.x;
    for (size_t base = (size_t)blockIdx.x * 64; base < n; base += (size_t)gridDim.x * 64) {
        if (base + W.here() >= n) continue;
        if constexpr (GUARD != 0) {
            const bool ok = ((int)E::limbs_ok(Pb, ld, base + W.here()) & (int)E::limbs_ok(Qb, ld, base + W.here())) != 0;
            if ((MA_WAVE_ALL(ok) != 0) != (GUARD > 0)) continue;
        }
        {
            spint ew[E::NW], fw[E::NW];
            load_be_record<typename E::P>(e, base + W.here(), ew);
            load_be_record<typename E::P>(f, base + W.here(), fw);
            E::jsf_digits(ew, fw, dj);
        }
        typename E::Point p, q, r;
        E::load(Pb, ld, base + W.here(), p);
        E::load(Qb, ld, base + W.here(), q);
        E::mul2_exact(dj, p, q, r, W);
        E::store(Rb, ld, base + W.here(), r);
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
