// modarith_amd/csrc/field.h -- per-thread field arithmetic for gfx950, one element per lane.
//
// Device-side counterpart of the functions modarith's generators emit into field.c
// (pseudo.py:223-1174 for 2^n-c primes, monty.py:352-1650 for Montgomery-form primes), written as
// C++ templates over a parameter struct P that the Python driver (modarith_amd/params.py ->
// csrc/generated/params_<PRIME>.h) emits.  Every limb stays in VGPRs: all loops are unrolled at
// compile time (static_for + if constexpr), so the shape of the prime (0, +-1, 2^k limbs) is
// resolved by the compiler exactly as the reference resolves it at generation time.
//
// Results are bit-identical to the reference's 64-bit field.c for EVERY input (also out-of-contract
// limbs): column sums are accumulated in a 128-bit integer with the same mask/shift points, and the
// 64-bit wrap-around spots (ma=a*mm, ta=2a, the scratch word s, q-v, v-1) are kept where the
// reference has them.  No MFMA: this is 64-bit integer carry-chain work.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace ma {

using spint = uint64_t;
using sspint = int64_t;
using dpint = unsigned __int128;

#ifndef MA_DEV          // tools/fe_host_check.hip defines it __host__ __device__ to run the same arithmetic on the CPU
#define MA_DEV __device__ __forceinline__
#endif

// MA_PIN(x): an empty, input-only asm that merely READS x.  A value with a second use is a leaf for LLVM's
// reassociation, so an accumulator pinned after every `acc += a * b` keeps the SOURCE order of a multiply-add chain:
// each step stays one v_mad_u64_u32 whose addend is the running value.  Without it the operands of a long sum are
// sorted by rank, every column starts at zero and the (late) carry of the previous column is added at the end with
// a separate 64-bit add (v_lshl_add_u64: the issue cost of a multiply-add) -- one extra instruction per column.
// The asm defines no register, so it emits nothing and does not trigger the gfx950 hazard rule "a VALU reading a
// register written by inline asm waits one state" (an `asm("" : "+v"(x))` pin costs an s_nop per step).
#if defined(__HIP_DEVICE_COMPILE__)
#define MA_PIN(x) asm volatile("" ::"v"(x))
#else
#define MA_PIN(x) ((void)0)
#endif

// lane_mask(c): all ones where c holds, 0 elsewhere, as a value the compiler knows NOTHING about.  `c ? 0 : w` on the words of a
// result invites it to move the whole computation of w under an EXEC region for the lanes that keep it, headed by
// s_cbranch_execz -- a branch on lane data (found by tools/ct_audit.py's EXEC-mask tracing in round 4: the zero-z records of
// k_fe_finish, the infinite results of the fused Weierstrass exports).  `w & lane_mask(!c)` is computed by every lane.
MA_DEV uint64_t lane_mask(bool c) {
    uint64_t m = c ? ~0ull : 0ull;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(m));
#endif
    return m;
}

template <int I, int End, class Fn>
MA_DEV void static_for(Fn&& fn) {
    if constexpr (I < End) {
        fn(std::integral_constant<int, I>{});
        static_for<I + 1, End>(fn);
    }
}

// 64x64 -> 128 product.  gfx950 has no 64-bit multiplier; the compiler lowers this to four
// v_mad_u64_u32 plus carries.
MA_DEV dpint mulw(spint a, spint b) { return (dpint)a * (dpint)b; }

// Column-sum policies.  Exact: every partial product is a full 64x64->128 multiply added into a 128-bit
// accumulator, identical to the reference for ALL inputs.  Split (FAST): operands are cut once at bit H into two
// 32-bit halves and the four half-products of a limb product go straight into three 64-bit accumulators
// (s0 + s1*2^H + s2*2^2H), one v_mad_u64_u32 each and no cross-word carries; the column value is the same
// integer, hence the same result, as long as every limb is below 2^(RADIX+2) -- which covers everything the
// reference's own functions produce or accept (tight, [p,2p) with the top limb unmasked, generic=False sums).
// The driver (emit.py) proves the accumulator bounds per prime and sets P::SPLIT (0 = not available).
template <bool FAST, int H>
struct Wide;
template <int H>
struct Wide<false, H> {
    using Opd = spint;
    static MA_DEV Opd prep(spint a) { return a; }
    template <unsigned long long D> static MA_DEV Opd prep_const() { return (spint)D; }
    struct Col {
        dpint t = 0;
        MA_DEV void mac(Opd a, Opd b) { t += (dpint)a * (dpint)b; }
        MA_DEV dpint sum() const { return t; }
    };
};
template <int H>
struct Wide<true, H> {
    struct Opd { uint32_t lo, hi; };
    static MA_DEV Opd prep(spint a) { return Opd{(uint32_t)a & ((1u << H) - 1u), (uint32_t)(a >> H)}; }
    // a compile-time constant as an operand whose non-zero halves sit in scalar registers the compiler cannot see through: a prime
    // limb 2^e written as a literal turns every "digit x limb" product of the Montgomery reduction into a 64-bit shift plus a
    // 64-bit add (two 64-bit instructions and a register pair per term) instead of ONE multiply-add (round 4: the scalar
    // multiplication over P-256 went from 242 to 150 VGPRs and 11 % up with the same change in monty_mul_half)
    template <unsigned long long D>
    static MA_DEV Opd prep_const() {
        uint32_t lo = (uint32_t)(D & ((1ull << H) - 1ull)), hi = (uint32_t)(D >> H);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MA_MHALF_SHIFT_TERMS)
        if constexpr ((D & ((1ull << H) - 1ull)) != 0) asm("" : "+s"(lo));
        if constexpr ((D >> H) != 0) asm("" : "+s"(hi));
#endif
        return Opd{lo, hi};
    }
    struct Col {
        uint64_t s0 = 0, s1 = 0, s2 = 0;
        MA_DEV void mac(Opd a, Opd b) {
            s0 += (uint64_t)a.lo * b.lo;
            s1 += (uint64_t)a.lo * b.hi;
            s1 += (uint64_t)a.hi * b.lo;
            s2 += (uint64_t)a.hi * b.hi;
        }
        MA_DEV dpint sum() const { return (dpint)s0 + ((dpint)s1 << H) + ((dpint)s2 << (2 * H)); }
    };
    // The running column value t of the product loops ("t += products; v = t & mask; t >>= Radix") kept as
    // t = c + s0 + s1*2^H + s2*2^2H and never assembled into 128 bits: c holds the previous column's t >> Radix
    // plus the one-word terms.  digit() returns t & mask and leaves t >> Radix in c using 64-bit operations
    // only; it needs Radix - H <= 32, Radix <= 2H < 64 and c + s0 + 2^Radix < 2^64, which the driver proves
    // per prime (emit.chain_ok -> P::CHAIN) under the same limb contract as SPLIT.  Same integers, same limbs.
    template <int R>
    struct Acc {
        uint64_t s0 = 0, s1 = 0, s2 = 0, c = 0;
        MA_DEV void mac(Opd a, Opd b) {
            s0 += (uint64_t)a.lo * b.lo;
            s1 += (uint64_t)a.lo * b.hi;
            s1 += (uint64_t)a.hi * b.lo;
            s2 += (uint64_t)a.hi * b.hi;
        }
        MA_DEV void add(uint64_t x) { c += x; }
        MA_DEV void add_twice(const Col& o) { s0 += o.s0 << 1; s1 += o.s1 << 1; s2 += o.s2 << 1; }
        MA_DEV uint64_t low() const { return c + s0 + (s1 << H) + (s2 << (2 * H)); }   // t mod 2^64
        MA_DEV uint64_t digit() {
            constexpr int L = R - H;
            const uint64_t lo = s0 + ((uint64_t)((uint32_t)s1 & (uint32_t)(((uint64_t)1 << L) - 1u)) << H) + c;
            c = (lo >> R) + (s1 >> L) + (s2 << (2 * H - R));
            s0 = s1 = s2 = 0;
            return lo & (((uint64_t)1 << R) - 1u);
        }
    };
};

// P::SPLIT_SPARSE (emit.split_is_sparse): the cut position SPLIT of this prime is proven from the number of prime limbs that actually enter
// the 64-bit accumulators of monty_mul / monty_reduce (powers of two, 0 and +-1 never do) -- NOT for product forms whose columns hold
// the dense 2N terms.  Parameter structs written before round 5 have no such member: false.
template <class P, class = void> struct split_sparse_of : std::false_type {};
template <class P> struct split_sparse_of<P, std::void_t<decltype(P::SPLIT_SPARSE)>> : std::bool_constant<P::SPLIT_SPARSE> {};

// PIN_: keep the multiply-add chains of the half-limb products pinned (MA_PIN).  On by default; the progenitor chain
// x^PE -- a separate, non-inlined function, for which no launch bound caps the register allocation -- is built from
// the unpinned variant: with pins that one function takes 248 VGPRs (the whole kernel then runs at one wave per
// SIMD), without them 132, and the extra 64-bit add per column costs less than the lost occupancy.
template <class P, bool FAST_ = false, bool PIN_ = true>
struct Field {
    template <class T> static MA_DEV void pin(T& x) { if constexpr (PIN_) MA_PIN(x); }
    static constexpr int N = P::N;
    // resident form of an element in the curve layer (csrc/curve.h): here the limbs themselves (csrc/fh51.h: half limbs)
    using limb_t = spint;
    static constexpr int NL = P::N;
    static MA_DEV void from_limbs(const spint* a, spint* h) { static_for<0, P::N>([&](auto I) { h[I] = a[I]; }); }
    static MA_DEV void to_limbs(const spint* h, spint* a) { static_for<0, P::N>([&](auto I) { a[I] = h[I]; }); }
    static MA_DEV spint pack(const spint* h, int k) { return h[k]; }
    static MA_DEV void unpack(spint w, spint* h, int k) { h[k] = w; }
    static constexpr int RADIX = P::RADIX;
    static constexpr spint Q = (spint)1 << RADIX;
    static constexpr spint MASK = Q - 1;
    using ov_t = std::conditional_t<P::BAD_OVERFLOW, dpint, spint>;     // see pm_modmul
    // (FOLD52: the 5 x 52 pseudo-Mersenne primes with a small mm -- 2^256-189 -- have no provable split form, mm * a does not
    // fit the three-accumulator scheme, but the half-limb columns with digit folding below do: pm_mul_half_ov)
    static constexpr bool FOLD52 = FAST_ && !P::MONTGOMERY && P::EPM && !P::OVERFLOW && P::RADIX == 52 && P::N == 5 && P::MM < (1ull << 16);
    static constexpr bool FAST = FAST_ && (P::SPLIT > 0 || FOLD52);
    using W = Wide<FAST, (P::SPLIT > 0 ? P::SPLIT : 32)>;
    using Opd = typename W::Opd;
    using Col = typename W::Col;
    static constexpr bool CHAINED = FAST && P::CHAIN;    // product loops on the 64-bit column chain (Wide::Acc)
    static constexpr bool SPLIT_SPARSE = split_sparse_of<P>::value;

    // ---------------------------------------------------------------- carries / normalisation
    // pseudo.py:223-251, monty.py:352-380 (arithmetic-shift form)
    static MA_DEV spint prop(spint* n) {
        sspint carry = (sspint)n[0];
        carry >>= RADIX;
        n[0] &= MASK;
        static_for<1, N - 1>([&](auto I) {
            constexpr int i = I;
            carry += (sspint)n[i];
            n[i] = (spint)carry & MASK;
            carry >>= RADIX;
        });
        n[N - 1] += (spint)carry;
        return (spint)0 - ((n[N - 1] >> 1) >> 62);
    }

    // n += x*p under an AND selector (caddp/addp: pseudo.py:202-213, monty.py:301-332)
    template <unsigned X>
    static MA_DEV void addp(spint* n, spint sel) {
        static_for<0, P::PP_CNT>([&](auto K) {
            constexpr int k = K;
            constexpr spint w = P::pp_val(k) * (spint)X;
            if constexpr (P::pp_sgn(k) < 0) n[P::pp_idx(k)] -= w & sel;
            else n[P::pp_idx(k)] += w & sel;
        });
    }
    // n -= x*p (subp: pseudo.py:216-220, monty.py:335-349)
    template <unsigned X>
    static MA_DEV void subp(spint* n) {
        static_for<0, P::PP_CNT>([&](auto K) {
            constexpr int k = K;
            constexpr spint w = P::pp_val(k) * (spint)X;
            if constexpr (P::pp_sgn(k) < 0) n[P::pp_idx(k)] += w;
            else n[P::pp_idx(k)] -= w;
        });
    }

    // pseudo.py:255-269
    static MA_DEV spint flatten(spint* n) {
        spint carry = prop(n);
        addp<1>(n, carry);
        (void)prop(n);
        return carry & 1;
    }
    // pseudo.py:272-283
    static MA_DEV spint modfsb(spint* n) {
        subp<1>(n);
        return flatten(n);
    }

    // ---------------------------------------------------------------- add / sub / neg
    // generic=True forms (pseudo.py:286-348, monty.py:417-490); out may alias in
    static MA_DEV void modadd(const spint* a, const spint* b, spint* n) {
        static_for<0, N>([&](auto I) { n[I] = a[I] + b[I]; });
        subp<2>(n);
        spint carry = prop(n);
        addp<2>(n, carry);
        (void)prop(n);
    }
    static MA_DEV void modsub(const spint* a, const spint* b, spint* n) {
        static_for<0, N>([&](auto I) { n[I] = a[I] - b[I]; });
        spint carry = prop(n);
        addp<2>(n, carry);
        (void)prop(n);
    }
    static MA_DEV void modneg(const spint* b, spint* n) {
        static_for<0, N>([&](auto I) { n[I] = (spint)0 - b[I]; });
        spint carry = prop(n);
        addp<2>(n, carry);
        (void)prop(n);
    }
    // "_u" forms: the generic=True functions above WITHOUT their closing carry propagation.  The result is the same integer as the
    // full function's (same choice of +2p, made on the propagated sum), but limbs 0..N-2 are the digits of the sum with 2p added
    // limb-wise on top -- not yet re-normalised.  Every add-class function starts by propagating its limb-wise sum, and what it
    // returns depends only on the INTEGER its operands represent, so a value that is consumed by modadd / modsub / modneg alone
    // may be produced this way: the consumer returns exactly the limbs it would return for the normalised operand.  Never feed a
    // "_u" value to a multiplication, a comparison or memory (the products' results depend on the limb representation).  Used by
    // the curve formulas (edwards.h, weierstrass.h) for the intermediate sums that only feed further sums: one carry chain of two saved.
    static MA_DEV void modadd_u(const spint* a, const spint* b, spint* n) {
        static_for<0, N>([&](auto I) { n[I] = a[I] + b[I]; });
        subp<2>(n);
        spint carry = prop(n);
        addp<2>(n, carry);
    }
    static MA_DEV void modsub_u(const spint* a, const spint* b, spint* n) {
        static_for<0, N>([&](auto I) { n[I] = a[I] - b[I]; });
        spint carry = prop(n);
        addp<2>(n, carry);
    }
    static MA_DEV void modneg_u(const spint* b, spint* n) {
        static_for<0, N>([&](auto I) { n[I] = (spint)0 - b[I]; });
        spint carry = prop(n);
        addp<2>(n, carry);
    }
    // generic=False forms with mp=2 (rfc7748.c:20; pseudo.py:294-325, monty.py:426-489)
    static MA_DEV void modadd_lazy(const spint* a, const spint* b, spint* n) {
        static_for<0, N>([&](auto I) { n[I] = a[I] + b[I]; });
        (void)prop(n);
    }
    static MA_DEV void modsub_lazy(const spint* a, const spint* b, spint* n) {
        static_for<0, N>([&](auto I) { n[I] = a[I] - b[I]; });
        addp<2>(n, ~(spint)0);
        (void)prop(n);
    }
    static MA_DEV void modneg_lazy(const spint* b, spint* n) {
        static_for<0, N>([&](auto I) { n[I] = (spint)0 - b[I]; });
        addp<2>(n, ~(spint)0);
        (void)prop(n);
    }

    static MA_DEV void modcpy(const spint* a, spint* c) {
        static_for<0, N>([&](auto I) { c[I] = a[I]; });
    }

    // ================================================================ pseudo-Mersenne family
    // second reduction pass (pseudo.py:557-611)
    static MA_DEV void pm_second_pass(dpint t, spint* v, spint* c) {
        constexpr int XC = P::XCESS;
        spint s, carry;
        if constexpr (P::FRED) {
            spint ut = (spint)t;
            if constexpr (XC > 0) {
                ut = (ut << XC) + (v[N - 1] >> (RADIX - XC));
                v[N - 1] &= ((spint)1 << (RADIX - XC)) - 1;
            }
            if constexpr (P::M > 1) ut *= (spint)P::M;
            s = v[0] + (ut & MASK);
            c[0] = s & MASK;
            if constexpr (P::CARRY_ON) {
                ut = (s >> RADIX) + (ut >> RADIX);
                s = v[1] + (ut & MASK);
                c[1] = s & MASK;
            }
            carry = (s >> RADIX) + (ut >> RADIX);
        } else {
            dpint ut = t;
            if constexpr (XC > 0) {
                ut = (ut << XC) + (dpint)(v[N - 1] >> (RADIX - XC));
                v[N - 1] &= ((spint)1 << (RADIX - XC)) - 1;
            }
            if constexpr (P::M > 1) ut *= (dpint)P::M;
            s = v[0] + ((spint)ut & MASK);
            c[0] = s & MASK;
            if constexpr (P::CARRY_ON) {
                ut = (dpint)(s >> RADIX) + (ut >> RADIX);
                s = v[1] + ((spint)ut & MASK);
                c[1] = s & MASK;
            }
            carry = (s >> RADIX) + (spint)(ut >> RADIX);
        }
        constexpr int k = P::CARRY_ON ? 2 : 1;
        c[k] = v[k] + carry;
        static_for<k + 1, N>([&](auto I) { c[I] = v[I]; });
    }

    // Comba rows, high half folded by mm (pseudo.py:616-659, getZM 390-438; overflow=False only)
    static MA_DEV void pm_modmul(const spint* a, const spint* b, spint* c) {
        dpint t = 0;
        spint v[N];
        // OVERFLOW form: the carried high part of the previous row's folded sum (pseudo.py:407-420) -- one word, or a double
        // word in the bad_overflow_mul form (pseudo.py:368-372, 412-416, 644-647: 601-610-bit moduli at 64 bits)
        ov_t hi_ov = 0;
        Opd A[N], B[N], MA[N];
        static_for<0, N>([&](auto I) { A[I] = W::prep(a[I]); B[I] = W::prep(b[I]); });
        if constexpr (P::EPM) static_for<1, N>([&](auto I) { MA[I] = W::prep(a[I] * (spint)P::MM); });
        static_for<0, N>([&](auto ROW) {
            constexpr int row = ROW;
            Col col;
            if constexpr (P::EPM) {
                static_for<row + 1, N>([&](auto K) {
                    constexpr int k = K;
                    col.mac(MA[k], B[N + row - k]);
                });
            } else if constexpr (row < N - 1) {
                Col hi;
                static_for<row + 1, N>([&](auto K) {
                    constexpr int k = K;
                    hi.mac(A[k], B[N + row - k]);
                });
                dpint tt = hi.sum();
                if constexpr (P::OVERFLOW) {
                    // mm times the folded sum would not fit 128 bits: its low limb is folded now, the rest with the
                    // next row (bad_overflow_mul = False form; the driver refuses primes that need the other one)
                    const spint lo = (spint)tt & MASK;
                    if constexpr (row == 0) t += (dpint)lo * (dpint)P::MM;
                    else if constexpr (P::BAD_OVERFLOW) t += (hi_ov + (dpint)lo) * (dpint)P::MM;
                    else t += (dpint)(spint)(lo + hi_ov) * (dpint)P::MM;
                    hi_ov = (ov_t)(tt >> RADIX);
                } else {
                    tt *= (dpint)P::MM;
                    t += tt;
                }
            }
            static_for<0, row + 1>([&](auto K) {
                constexpr int k = K;
                col.mac(A[k], B[row - k]);
            });
            t += col.sum();
            if constexpr (P::OVERFLOW && row == N - 1) t += (dpint)hi_ov * (dpint)P::MM;     // pseudo.py:435-436
            v[row] = (spint)t & MASK;
            t >>= RADIX;
        });
        pm_second_pass(t, v, c);
    }

    // squaring rows (pseudo.py:663-702, getZS 441-554)
    static MA_DEV void pm_modsqr(const spint* a, spint* c) {
        dpint t = 0;
        spint v[N];
        ov_t hi_ov = 0;        // OVERFLOW form (pseudo.py:492-493, 536-550; double word in the bad_overflow_sqr form)
        Opd A[N], TA[N], MA[N];
        static_for<0, N>([&](auto I) { A[I] = W::prep(a[I]); });
        if constexpr (P::EPM) {
            static_for<1, N>([&](auto I) { TA[I] = W::prep(a[I] * (spint)2); });
            static_for<1, N>([&](auto I) { MA[I] = W::prep(a[I] * (spint)P::MM); });
        }
        static_for<0, N>([&](auto ROW) {
            constexpr int row = ROW;
            // folded (high) part: pairs (k, l) with k + l = N + row, row < k <= l < N
            constexpr int hk0 = row + 1, hpairs = (N - 1 - hk0 + 1) / 2;  // strict pairs k < l
            Col col;
            if constexpr (P::EPM) {
                static_for<0, hpairs>([&](auto J) {
                    constexpr int k = hk0 + J, l = N - 1 - J;
                    col.mac(MA[k], TA[l]);
                });
                if constexpr ((N - hk0) % 2 == 1) {
                    constexpr int k = hk0 + hpairs;
                    col.mac(MA[k], A[k]);
                }
            } else if constexpr (row < N - 1) {
                Col cross, sq;
                static_for<0, hpairs>([&](auto J) {
                    constexpr int k = hk0 + J, l = N - 1 - J;
                    cross.mac(A[k], A[l]);
                });
                dpint tt = cross.sum();
                if constexpr (hpairs > 0) tt *= 2;
                if constexpr ((N - hk0) % 2 == 1) {
                    constexpr int k = hk0 + hpairs;
                    sq.mac(A[k], A[k]);
                    tt += sq.sum();
                }
                if constexpr (P::OVERFLOW) {
                    const spint lo = (spint)tt & MASK;
                    if constexpr (row == 0) t += (dpint)lo * (dpint)P::MM;
                    else if constexpr (P::BAD_OVERFLOW) t += (hi_ov + (dpint)lo) * (dpint)P::MM;
                    else t += (dpint)(spint)(lo + hi_ov) * (dpint)P::MM;
                    hi_ov = (ov_t)(tt >> RADIX);
                } else {
                    tt *= (dpint)P::MM;
                    t += tt;
                }
            } else if constexpr (P::OVERFLOW) {
                t += (dpint)hi_ov * (dpint)P::MM;          // row N-1 (pseudo.py:537-538)
            }
            // low part: pairs (k, l) with k + l = row
            constexpr int lpairs = (row + 1) / 2;
            if constexpr (P::EPM) {
                static_for<0, lpairs>([&](auto J) {
                    constexpr int k = J, l = row - J;
                    col.mac(A[k], TA[l]);
                });
                if constexpr (row % 2 == 0) col.mac(A[row / 2], A[row / 2]);
                t += col.sum();
            } else {
                Col cross, sq;
                static_for<0, lpairs>([&](auto J) {
                    constexpr int k = J, l = row - J;
                    cross.mac(A[k], A[l]);
                });
                dpint t2 = cross.sum();
                if constexpr (lpairs > 0) t2 *= 2;
                if constexpr (row % 2 == 0) {
                    sq.mac(A[row / 2], A[row / 2]);
                    t2 += sq.sum();
                }
                t += t2;
            }
            v[row] = (spint)t & MASK;
            t >>= RADIX;
        });
        pm_second_pass(t, v, c);
    }

    // ---------------------------------------------------------------- four-accumulator split (EPM primes without a provable SPLIT)
    // 2^521 - 1 in nine 58-bit limbs has no three-accumulator split: with limbs below 2^60 and the premultiplied operand mm * a
    // one bit wider, eighteen cross products of 2^60 pass 2^64 by a sixth of a bit (emit.split_point), and the exact rows --
    // 128-bit sums through the carry-out of v_mad_u64_u32 -- compile into twice as many moves, carry adds and hazard nops as
    // multiply-adds (round 4: 19 % of k_ed_mul<NIST521> were multiplier instructions).  Two changes make the split fit: the cross
    // products lo x hi and hi x lo go to separate accumulators (nine terms of 2^60 each: below 2^63.2), and the factor mm is applied
    // to the folded SUM instead of to every operand (mm * sum(a_k b_l) is the integer sum(mm a_k * b_l)).  Same column integers as
    // pm_modmul / pm_modsqr, hence the same limbs.  Cut at H4 = (Radix + 2) / 2: both halves below 2^H4.
    static constexpr int H4 = (RADIX + 2) / 2;
    static constexpr bool SPLIT4 = FAST_ && !FAST && !P::MONTGOMERY && P::EPM && !P::OVERFLOW && !P::BAD_OVERFLOW && (RADIX + 2) % 2 == 0 && H4 <= 31 &&
                                   ((unsigned long long)N << (2 * H4 - 32)) < (1ull << 32) && P::MM <= 16;
    struct Col4 {
        uint64_t s0 = 0, s1a = 0, s1b = 0, s2 = 0;
        MA_DEV void mac(uint32_t al, uint32_t ah, uint32_t bl, uint32_t bh) {
            s0 += (uint64_t)al * bl;
            s1a += (uint64_t)al * bh;
            s1b += (uint64_t)ah * bl;
            s2 += (uint64_t)ah * bh;
        }
        MA_DEV dpint sum() const { return (dpint)s0 + (((dpint)s1a + (dpint)s1b) << H4) + ((dpint)s2 << (2 * H4)); }
    };
    static MA_DEV void split4(const spint* a, uint32_t* lo, uint32_t* hi) {
        static_for<0, N>([&](auto I) {
            lo[I] = (uint32_t)a[I] & ((1u << H4) - 1u);
            hi[I] = (uint32_t)(a[I] >> H4);
        });
    }
    static MA_DEV void pm_modmul_split4(const spint* a, const spint* b, spint* c) {
        uint32_t al[N], ah[N], bl[N], bh[N];
        split4(a, al, ah);
        split4(b, bl, bh);
        dpint t = 0;
        spint v[N];
        static_for<0, N>([&](auto ROW) {
            constexpr int row = ROW;
            Col4 hi, lo;
            static_for<row + 1, N>([&](auto K) {
                constexpr int k = K;
                hi.mac(al[k], ah[k], bl[N + row - k], bh[N + row - k]);
            });
            static_for<0, row + 1>([&](auto K) {
                constexpr int k = K;
                lo.mac(al[k], ah[k], bl[row - k], bh[row - k]);
            });
            if constexpr (row < N - 1) t += hi.sum() * (dpint)P::MM;
            t += lo.sum();
            v[row] = (spint)t & MASK;
            t >>= RADIX;
        });
        pm_second_pass(t, v, c);
    }
    static MA_DEV void pm_modsqr_split4(const spint* a, spint* c) {
        uint32_t al[N], ah[N];
        split4(a, al, ah);
        dpint t = 0;
        spint v[N];
        static_for<0, N>([&](auto ROW) {
            constexpr int row = ROW;
            // folded (high) part: pairs (k, l), k + l = N + row, row < k <= l < N;  low part: pairs with k + l = row
            constexpr int hk0 = row + 1, hpairs = (N - 1 - hk0 + 1) / 2, lpairs = (row + 1) / 2;
            Col4 hcross, lcross, hsq, lsq;
            static_for<0, hpairs>([&](auto J) {
                constexpr int k = hk0 + J, l = N - 1 - J;
                hcross.mac(al[k], ah[k], al[l], ah[l]);
            });
            static_for<0, lpairs>([&](auto J) {
                constexpr int k = J, l = row - J;
                lcross.mac(al[k], ah[k], al[l], ah[l]);
            });
            dpint h = hcross.sum() * 2, l = lcross.sum() * 2;
            if constexpr ((N - hk0) % 2 == 1) {
                constexpr int k = hk0 + hpairs;
                hsq.mac(al[k], ah[k], al[k], ah[k]);
                h += hsq.sum();
            }
            if constexpr (row % 2 == 0) {
                lsq.mac(al[row / 2], ah[row / 2], al[row / 2], ah[row / 2]);
                l += lsq.sum();
            }
            if constexpr (row < N - 1) t += h * (dpint)P::MM;
            t += l;
            v[row] = (spint)t & MASK;
            t >>= RADIX;
        });
        pm_second_pass(t, v, c);
    }

    // the same rows on the 64-bit column chain (EPM primes; see Wide::Acc)
    static MA_DEV void pm_modmul_chain(const spint* a, const spint* b, spint* c) {
        static_assert(P::EPM && !P::OVERFLOW, "chained products are built for the EPM form only");
        typename W::template Acc<RADIX> t;
        spint v[N];
        Opd A[N], B[N], MA[N];
        static_for<0, N>([&](auto I) { A[I] = W::prep(a[I]); B[I] = W::prep(b[I]); });
        static_for<1, N>([&](auto I) { MA[I] = W::prep(a[I] * (spint)P::MM); });
        static_for<0, N>([&](auto ROW) {
            constexpr int row = ROW;
            static_for<row + 1, N>([&](auto K) {
                constexpr int k = K;
                t.mac(MA[k], B[N + row - k]);
            });
            static_for<0, row + 1>([&](auto K) {
                constexpr int k = K;
                t.mac(A[k], B[row - k]);
            });
            v[row] = t.digit();
        });
        pm_second_pass((dpint)t.c, v, c);
    }
    static MA_DEV void pm_modsqr_chain(const spint* a, spint* c) {
        static_assert(P::EPM && !P::OVERFLOW, "chained products are built for the EPM form only");
        typename W::template Acc<RADIX> t;
        spint v[N];
        Opd A[N], TA[N], MA[N];
        static_for<0, N>([&](auto I) { A[I] = W::prep(a[I]); });
        static_for<1, N>([&](auto I) { TA[I] = W::prep(a[I] * (spint)2); });
        static_for<1, N>([&](auto I) { MA[I] = W::prep(a[I] * (spint)P::MM); });
        static_for<0, N>([&](auto ROW) {
            constexpr int row = ROW;
            constexpr int hk0 = row + 1, hpairs = (N - 1 - hk0 + 1) / 2;
            static_for<0, hpairs>([&](auto J) {
                constexpr int k = hk0 + J, l = N - 1 - J;
                t.mac(MA[k], TA[l]);
            });
            if constexpr ((N - hk0) % 2 == 1) {
                constexpr int k = hk0 + hpairs;
                t.mac(MA[k], A[k]);
            }
            constexpr int lpairs = (row + 1) / 2;
            static_for<0, lpairs>([&](auto J) {
                constexpr int k = J, l = row - J;
                t.mac(A[k], TA[l]);
            });
            if constexpr (row % 2 == 0) t.mac(A[row / 2], A[row / 2]);
            v[row] = t.digit();
        });
        pm_second_pass((dpint)t.c, v, c);
    }

    // ---------------------------------------------------------------- half-limb columns (2^255-19 shape)
    // For an odd radix R = 2H - 1 and p = 2^(N R) - MM the product loops above compute the integer
    //     T = sum_{k,j} a_k b_j w(k+j),   w(m) = 2^(R m) for m < N,  MM 2^(R (m-N)) for m >= N,
    // emit its base-2^R digits v_0 .. v_{N-1} and hand T >> (N R) to the second pass.  T is bilinear in the limbs, so
    // it can be accumulated from HALF limbs a_k = lo_k + 2^H hi_k (lo < 2^H, hi < 2^(R+2-H)) just as well: half-limb
    // i sits at bit R (i/2) + H (i%2), two odd halves meet one bit above a column boundary (factor 2), and the 2N
    // column sums h_i are exactly the columns of the 2N x (H or H-1)-bit mixed radix -- what csrc/fe26.h does for the
    // ladder.  Every partial product is ONE v_mad_u64_u32 into a 64-bit column (100 per modmul instead of 100 + the
    // three-accumulator digit() assembly; 174 instead of 273 instructions), the carry chain yields the 2N half digits,
    // v_r = t_{2r} + 2^H t_{2r+1} are the reference's digits and the last carry is its T >> (N R): the SAME limbs as
    // pm_modmul for every input inside the limb contract (limbs < 2^(R+2)).
    // One family of terms needs care: two ODD halves with i + j = 2N exactly (limbs k + j = N - 1) weigh
    // 2^(2H) 2^(R (N-1)) = 2 * 2^(N R) -- congruent to 2 MM, but the reference keeps a_k b_j whole in row N-1, so in
    // its T they sit ABOVE bit N R.  Wrapping them into column 0 would give T - 2 p X: the same residue, but a
    // different integer, hence (whenever the low part borrows) other digits.  They are accumulated on top of the last
    // carry instead, so that the value handed to the second pass is the reference's t.
    // Bounds for R = 51, H = 26, N = 5, MM = 19, limbs < 2^53: lo < 2^26, hi < 2^27, 19 hi < 2^31.3; the largest
    // column (k = 2: four wrapped odd-odd terms 2 * 19 * hi hi, three wrapped even-even ones, three plain) stays below
    // 4 * 2^59.3 + 2^59 < 2^61.6, carries below 2^36, the value handed to the second pass below 2^36 + 5 * 2^55.
    static constexpr bool HALF = FAST && !P::MONTGOMERY && P::EPM && !P::OVERFLOW && P::FRED && RADIX == 51 && N == 5 && P::MM == 19;
    static constexpr int HH = (RADIX + 1) / 2;                      // 26
    static constexpr int hbits(int i) { return (i & 1) ? RADIX - HH : HH; }
    static MA_DEV void half_split(const spint* a, uint32_t* f) {
        static_for<0, N>([&](auto K) {
            constexpr int k = K;
            f[2 * k] = (uint32_t)a[k] & ((1u << HH) - 1u);
            f[2 * k + 1] = (uint32_t)(a[k] >> HH);
        });
    }
    // half digits t[0..2N) and the final carry -> the reference's limbs (second pass included)
    static MA_DEV void half_finish(const uint32_t* t, uint64_t carry, spint* c) {
        spint v[N];
        static_for<0, N>([&](auto K) {
            constexpr int k = K;
            v[k] = (spint)t[2 * k] | ((spint)t[2 * k + 1] << HH);
        });
        pm_second_pass((dpint)carry, v, c);
    }
    static MA_DEV void pm_modmul_half(const spint* a, const spint* b, spint* c) {
        constexpr int M = 2 * N;
        uint32_t f[M], g[M], g19[M], f2[M], t[M];
        half_split(a, f);
        half_split(b, g);
        static_for<1, M>([&](auto J) { g19[J] = (uint32_t)P::MM * g[J]; });
        static_for<0, N>([&](auto K) { f2[2 * K + 1] = 2u * f[2 * K + 1]; });
        uint64_t cy = 0;
        static_for<0, M>([&](auto KK) {
            constexpr int k = KK;
            uint64_t acc = cy;
            static_for<0, M>([&](auto II) {
                constexpr int i = II;
                constexpr int j = (k - i + M) % M;
                constexpr bool wrp = (i + j) >= M;
                constexpr bool dbl = (i & 1) && (j & 1);
                if constexpr (!(dbl && i + j == M)) {             // (those go on top of the last carry, below)
                    const uint32_t x = dbl ? f2[i] : f[i];
                    const uint32_t y = wrp ? g19[j] : g[j];
                    acc += (uint64_t)x * y;
                    pin(acc);
                }
            });
            t[k] = (uint32_t)acc & ((1u << hbits(k)) - 1u);
            cy = acc >> hbits(k);
        });
        static_for<0, N>([&](auto K) {
            constexpr int i = 2 * K + 1;
            cy += (uint64_t)f2[i] * g[M - i];
            pin(cy);
        });
        half_finish(t, cy, c);
    }
    static MA_DEV void pm_modsqr_half(const spint* a, spint* c) {
        constexpr int M = 2 * N;
        uint32_t f[M], f2[M], f4[M], f19[M], t[M];
        half_split(a, f);
        static_for<0, M>([&](auto I) { f2[I] = 2u * f[I]; });
        static_for<0, N>([&](auto K) { f4[2 * K + 1] = 4u * f[2 * K + 1]; });
        static_for<1, M>([&](auto J) { f19[J] = (uint32_t)P::MM * f[J]; });
        uint64_t cy = 0;
        static_for<0, M>([&](auto KK) {
            constexpr int k = KK;
            uint64_t acc = cy;
            static_for<0, M>([&](auto II) {
                constexpr int i = II;
                constexpr int j = (k - i + M) % M;
                if constexpr (i <= j) {
                    constexpr bool wrp = (i + j) >= M;
                    constexpr bool odd2 = (i & 1) && (j & 1);
                    uint32_t x, y;
                    if constexpr (i == j) {                       // f_i^2, times 2 if odd, times MM if wrapped
                        x = odd2 ? f2[i] : f[i];
                        y = wrp ? f19[j] : f[j];
                    } else {                                      // 2 f_i f_j, times 2 if both odd, times MM if wrapped
                        x = odd2 ? f4[i] : f2[i];
                        y = wrp ? f19[j] : f[j];
                    }
                    if constexpr (!(odd2 && i + j == M)) {
                        acc += (uint64_t)x * y;
                        pin(acc);
                    }
                }
            });
            t[k] = (uint32_t)acc & ((1u << hbits(k)) - 1u);
            cy = acc >> hbits(k);
        });
        static_for<0, N>([&](auto K) {                            // odd i + j = 2N: on top of the last carry
            constexpr int i = 2 * K + 1, j = M - i;
            if constexpr (i < j) { cy += (uint64_t)f4[i] * f[j]; pin(cy); }
            else if constexpr (i == j) { cy += (uint64_t)f2[i] * f[j]; pin(cy); }
        });
        half_finish(t, cy, c);
    }

    // The "overflow" pseudo-Mersenne form (secp256k1 as pseudo.py builds it at 64 bits: R = 52 = 2*26, N = 5, 2^260 = mm with
    // mm = 2^36 + 0x3d10): the reference folds the high rows by lo/hi_ov pieces (pm_modmul above), which sums to exactly
    // T = L + mm * HI with L / HI the low / high halves of the limb product.  In radix 2^26: the high half columns
    // 10..18 (without the odd-odd pairs of column 10, which belong to limb row N-1 and stay above bit N R, cf.
    // pm_modmul_half) are carried into 26-bit digits d_m, each digit folds into column m as d_m * 0x3d10 and into column
    // m+1 as d_m << 10, columns >= 10 are the value handed to the second pass.  100 + 11 multiply-adds.
    // Bounds (limbs < 2^54: lo < 2^26, hi < 2^28): a half column holds at most 10 products below 2^56 and fold terms
    // below 2^41: < 2^60; the second-pass value stays below 2^46.
    // The same routine serves FOLD52 (mm < 2^16, EPM form: mm * a_k times b_j sums to the same T = L + mm * HI), with the
    // 2^36 part absent.  There the reference itself wraps mm * a_k at 64 bits, so "inside the contract" means limbs below
    // 2^64 / mm (2^52.4 for mm = 0xbd0) -- true of everything the field functions return (masked digits plus a small
    // carry), which is all the curve layer ever multiplies.
    static constexpr bool HALF_OV = FAST && !P::MONTGOMERY && RADIX == 52 && N == 5 &&
                                    ((!P::EPM && P::OVERFLOW && (P::MM >> 36) == 1 && (P::MM & 0xfffffffffull) < (1ull << 16)) || FOLD52);
    template <bool SQR>
    static MA_DEV void pm_mul_half_ov(const spint* a, const spint* b, spint* c) {
        constexpr int H = 26, M = 2 * N;
        constexpr uint32_t HM = (1u << H) - 1u;
        constexpr uint32_t MLO = (uint32_t)(P::MM & 0xfffffffffull);         // mm = MHI * 2^36 + MLO
        constexpr bool MHI = (P::MM >> 36) != 0;
        uint32_t f[M], g[M], f2[M], u[M], d[M + 1];
        static_for<0, N>([&](auto K) {
            constexpr int k = K;
            f[2 * k] = (uint32_t)a[k] & HM;
            f[2 * k + 1] = (uint32_t)(a[k] >> H);
            if constexpr (!SQR) {
                g[2 * k] = (uint32_t)b[k] & HM;
                g[2 * k + 1] = (uint32_t)(b[k] >> H);
            }
        });
        if constexpr (SQR) static_for<0, M>([&](auto I) { f2[I] = 2u * f[I]; });
        // one half column of the product; TOP selects the odd-odd pairs of column M (limb row N-1), WRAP the others
        auto column = [&](auto KK, uint64_t acc, auto only_top, auto skip_top) -> uint64_t {
            constexpr int k = KK;
            constexpr int lo = k < M ? 0 : k - (M - 1), hi = k < M ? k : M - 1;
            static_for<lo, hi + 1>([&](auto II) {
                constexpr int i = II, j = k - i;
                constexpr bool top = (k == M) && (i & 1) && (j & 1);
                if constexpr ((!decltype(only_top)::value || top) && !(decltype(skip_top)::value && top)) {
                    if constexpr (!SQR) {
                        acc += (uint64_t)f[i] * g[j];
                        pin(acc);
                    } else if constexpr (i <= j) {
                        acc += (uint64_t)((i < j) ? f2[i] : f[i]) * f[j];
                        pin(acc);
                    }
                }
            });
            return acc;
        };
        // high part: half columns M .. 2M-2 -> digits d_0 .. d_8, then the rest d_9, d_10
        uint64_t hcy = 0;
        static_for<0, M - 1>([&](auto MM_) {
            constexpr int m = MM_;
            const uint64_t acc = column(std::integral_constant<int, M + m>{}, hcy, std::false_type{}, std::true_type{});
            d[m] = (uint32_t)acc & HM;
            hcy = acc >> H;
        });
        d[M - 1] = (uint32_t)hcy & HM;
        d[M] = (uint32_t)(hcy >> H);
        // low part + folded digits
        uint64_t cy = 0;
        static_for<0, M>([&](auto KK) {
            constexpr int k = KK;
            uint64_t acc = column(KK, cy, std::false_type{}, std::false_type{});
            acc += (uint64_t)d[k] * MLO;
            pin(acc);
            if constexpr (k > 0 && MHI) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MA_MHALF_SHIFT_TERMS)
                uint32_t p10 = 1u << 10;                     // (one multiply-add instead of a 64-bit shift and a 64-bit add: see monty_mul_half)
                asm("" : "+s"(p10));
                acc += (uint64_t)d[k - 1] * p10;
#else
                acc += (uint64_t)d[k - 1] << 10;
#endif
                pin(acc);
            }
            u[k] = (uint32_t)acc & HM;
            cy = acc >> H;
        });
        // what lies at or above bit N R: the running carry, the folds of d_9 / d_10 and the odd-odd pairs of column M
        uint64_t t = column(std::integral_constant<int, M>{}, cy, std::true_type{}, std::false_type{});
        t += (uint64_t)d[M] * MLO;
        if constexpr (MHI) t += ((uint64_t)d[M - 1] << 10) + ((uint64_t)d[M] << (10 + H));
        spint v[N];
        static_for<0, N>([&](auto K) {
            constexpr int k = K;
            v[k] = (spint)u[2 * k] | ((spint)u[2 * k + 1] << H);
        });
        pm_second_pass((dpint)t, v, c);
    }

    // pseudo.py:705-728; (dpint)b sign-extends a negative int exactly as the emitted C does
    static MA_DEV void pm_modmli(const spint* a, int b, spint* c) {
        dpint t = 0;
        spint v[N];
        const dpint bw = (dpint)(__int128)b;
        static_for<0, N>([&](auto I) {
            t += (dpint)a[I] * bw;
            v[I] = (spint)t & MASK;
            t >>= RADIX;
        });
        pm_second_pass(t, v, c);
    }

    // ================================================================ Montgomery family (ndash == 1)
    // Reduction contribution to column c: digit v_j meets signed prime limb l = c - j, 1 <= l <= LMAX
    // (mul_process with the gone_neg / mask_set borrow convention: monty.py:597-627, 717-738, 778-838).
    static constexpr int LMAX = P::MONTGOMERY ? (P::E ? N : N - 1) : 0;  // highest prime-limb index
    static constexpr int JMAX = LMAX;                                   // highest digit index

    // monty.py's PM form (an exploitable pseudo-Mersenne 2^n - M given to monty.py: low prime limb -M, monty.py:284-288, 700-870):
    // the low limb is carried with the borrow technique scaled by M -- column 0 adds M*(q - v0), every later column adds
    // M*mask up front and takes M*v_i back with its own digit, and the last limb gives back M (or v_N - M below a virtual limb).
    static constexpr spint PM_M = (spint)P::PM_M;
    template <int C>
    static MA_DEV void monty_reduce(dpint& t, Col& col, const spint* v, const Opd* V) {
        if constexpr (PM_M != 0 && C >= 1) t += (dpint)(spint)(PM_M * MASK);
        constexpr int NEG = P::NEG_LIMB;       // index (>=1) of the single -1 limb, or 0 if none
        constexpr bool scratch = (NEG > 0) && (C > NEG);  // gone_neg at column start -> s = mask
        spint s = MASK;
        static_for<1, LMAX + 1>([&](auto L) {
            constexpr int l = L;
            constexpr int j = C - l;
            if constexpr (j >= 0 && j <= JMAX && j < C) {
                constexpr long long d = P::ppw(l);
                if constexpr (d > 1) {
                    if constexpr ((d & (d - 1)) == 0) {
                        constexpr int e = __builtin_ctzll((unsigned long long)d);
                        t += (dpint)v[j] << e;
                    } else {
                        col.mac(V[j], W::prep((spint)d));
                    }
                } else if constexpr (d == 1) {
                    if constexpr (scratch) s += v[j]; else t += (dpint)v[j];
                } else if constexpr (d == -1) {
                    if constexpr (scratch) s -= v[j];
                    else t += (dpint)(spint)(Q - v[j]);   // first negative use (C == NEG, j == 0)
                } else {
                    static_assert(d == 0, "negative prime limbs other than -1 are not supported");
                }
            }
        });
        if constexpr (scratch) t += (dpint)s;
    }

    // reduction digit of a column (monty.py:698-712, 740-751): t & mask when ndash == 1 ("Montgomery
    // friendly"), else (t*ndash) & mask followed by t += v*p0 so that the low RADIX bits cancel
    template <int COL>
    static MA_DEV spint monty_digit(dpint& t) {
        if constexpr (P::NDASH == 1) {
            return (spint)t & MASK;
        } else if constexpr (PM_M != 0) {
            const spint v = ((spint)t * (spint)P::NDASH) & MASK;
            if constexpr (COL == 0) t += (dpint)(spint)(PM_M * (Q - v));
            else t -= (dpint)(spint)(PM_M * v);
            return v;
        } else {
            static_assert(P::ppw(0) > 0, "full Montgomery reduction expects a positive low prime limb");
            const spint v = ((spint)t * (spint)P::NDASH) & MASK;
            if constexpr (P::ppw(0) == 1) {
                t += (dpint)v;
            } else {
                Col c0;
                c0.mac(W::prep(v), W::prep((spint)P::ppw(0)));
                t += c0.sum();
            }
            return v;
        }
    }

    template <bool SQR>
    static MA_DEV void monty_mul(const spint* a, const spint* b, spint* c) {
        constexpr int NCOL = P::E ? 2 * N : 2 * N - 1;
        dpint t = 0;
        spint v[JMAX + 1];
        Opd V[JMAX + 1];
        Opd A[N], B[N];
        static_for<0, N>([&](auto I) { A[I] = W::prep(a[I]); });
        if constexpr (!SQR) static_for<0, N>([&](auto I) { B[I] = W::prep(b[I]); });
        static_for<0, NCOL>([&](auto CC) {
            constexpr int col = CC;
            constexpr int lo = col < N ? 0 : col - (N - 1);
            constexpr int hi = col < N ? col : N - 1;
            Col acc;
            if constexpr (lo <= hi) {
                if constexpr (!SQR) {
                    // getZMU / getZMD (monty.py:493-537)
                    static_for<lo, hi + 1>([&](auto K) {
                        constexpr int k = K;
                        acc.mac(A[k], B[col - k]);
                    });
                } else {
                    // getZSU / getZSD (monty.py:540-590): tot = 2*sum(cross) + square
                    constexpr int pairs = (hi - lo + 1) / 2;
                    Col cross;
                    static_for<0, pairs>([&](auto J) {
                        constexpr int k = lo + J;
                        cross.mac(A[k], A[col - k]);
                    });
                    dpint tot = cross.sum();
                    if constexpr (pairs > 0) tot *= 2;
                    t += tot;
                    if constexpr (col % 2 == 0) acc.mac(A[col / 2], A[col / 2]);
                }
            }
            monty_reduce<col>(t, acc, v, V);
            t += acc.sum();
            if constexpr (col <= JMAX) {
                v[col] = monty_digit<col>(t);
                V[col] = W::prep(v[col]);
            } else {
                c[col - JMAX - 1] = (spint)t & MASK;
            }
            t >>= RADIX;
        });
        if constexpr (P::E) {
            if constexpr (PM_M != 0) t += (dpint)(spint)(v[N] - PM_M);
            else if constexpr (P::NEG_LIMB > 0) t += (dpint)(spint)(v[N] - (spint)1);
            else t += (dpint)v[N];
        } else {
            if constexpr (PM_M != 0) t -= (dpint)PM_M;
            else if constexpr (P::NEG_LIMB > 0) t -= (dpint)1;
        }
        c[N - 1] = (spint)t;
    }

    // ---------------------------------------------------------------- half-limb columns, Montgomery shape (P-256)
    // The same idea as pm_modmul_half for an EVEN radix R = 2H with "Montgomery-friendly" digits (ndash == 1,
    // p = -1 mod 2^R, no virtual limb, no negative limb above limb 0): the reference's column loop is
    //     t += column C of a*b + sum_l v_{C-l} ppw(l);   v_C = t mod 2^R (C <= JMAX) or c_{C-N} = t mod 2^R;   t >>= R
    // and ends with c_{N-1} = t.  Everything is an integer sum in which each term has a definite weight, so it can be
    // accumulated from half limbs in radix 2^H: a_k = lo + 2^H hi puts lo lo' / cross / hi hi' at half columns
    // 2(k+j), +1, +2 (2H = R: no doubling factors), a digit v_j = u_2j + 2^H u_2j+1 times a prime limb 2^e goes to half
    // column 2(j+l) + e/H as (u << e%H), times a general limb d = d_lo + 2^H d_hi as four multiply-adds, and the low
    // limb -1 is the digit extraction itself.  Nothing wraps, so -- unlike the pseudo-Mersenne form -- there is no
    // early-fold subtlety: the half digits, two by two, ARE the reference's digits, and the running value after half
    // column 4N-3 is its last limb.  100 multiply-adds + 20 reduction multiply-adds and 20 shift-adds per modmul of
    // P-256 (about 210 instructions instead of 355).
    // Bounds for R = 52, N = 5, limbs < 2^54: lo < 2^26, hi < 2^28; a half column holds at most 10 products below 2^56,
    // digit terms below 2^49 and a carry below 2^38: < 2^60.
    static constexpr bool MHALF = FAST && P::MONTGOMERY && P::NDASH == 1 && !P::E && P::NEG_LIMB == 0 && RADIX == 52 && N == 5 &&
                                  P::ppw(0) == -1;
    template <bool SQR>
    static MA_DEV void monty_mul_half(const spint* a, const spint* b, spint* c) {
        constexpr int H = RADIX / 2, M = 2 * N;             // 26-bit halves, 10 of them per operand
        constexpr uint32_t HM = (1u << H) - 1u;
        uint32_t f[M], g[M], u[2 * M];
        static_for<0, N>([&](auto K) {
            constexpr int k = K;
            f[2 * k] = (uint32_t)a[k] & HM;
            f[2 * k + 1] = (uint32_t)(a[k] >> H);
            if constexpr (!SQR) {
                g[2 * k] = (uint32_t)b[k] & HM;
                g[2 * k + 1] = (uint32_t)(b[k] >> H);
            }
        });
        uint32_t f2[M];
        if constexpr (SQR) static_for<0, M>([&](auto I) { f2[I] = 2u * f[I]; });
        uint64_t cy = 0;
        // half columns 0 .. 2M-2 carry products; digits are produced up to half column 2N-1 (JMAX = N-1 limbs), the limbs
        // c_0 .. c_{N-2} are half columns 2N .. 4N-3, and what is left afterwards is c_{N-1}
        static_for<0, 4 * N - 1>([&](auto KK) {
            constexpr int k = KK;                            // half column; k = 4N-2 is the remainder (not masked)
            uint64_t acc = cy;
            constexpr int lo = k < M ? 0 : k - (M - 1), hi = k < M ? k : M - 1;
            static_for<lo, hi + 1>([&](auto II) {
                constexpr int i = II, j = k - i;
                if constexpr (!SQR) {
                    acc += (uint64_t)f[i] * g[j];
                    pin(acc);
                } else if constexpr (i <= j) {
                    acc += (uint64_t)((i < j) ? f2[i] : f[i]) * f[j];
                    pin(acc);
                }
            });
            // reduction terms: digit halves u[2j], u[2j+1] times prime limb l >= 1 land at bit R (j + l) + (bits of the limb)
            static_for<1, N>([&](auto L) {
                constexpr int l = L;
                constexpr long long d = P::ppw(l);
                static_assert(d >= 0, "only the low prime limb may be negative here");
                if constexpr (d > 0) {
                    if constexpr ((d & (d - 1)) == 0) {
                        constexpr int e = __builtin_ctzll((unsigned long long)d);
                        constexpr int q = e / H, sh = e % H;                 // u << sh at half column 2(j+l) + q + (half index)
                        static_for<0, M>([&](auto JJ) {                       // JJ = index of the digit half
                            constexpr int jh = JJ;
                            if constexpr (2 * l + q + jh == k && jh < k) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MA_MHALF_SHIFT_TERMS)
                                // u * 2^sh as ONE multiply-add (5 issue cycles) instead of a 64-bit shift and a 64-bit add (two
                                // 64-bit instructions, 9-10, and a register pair per shifted digit: the scalar multiplication over P-256
                                // went from 242 to 150 VGPRs with this line): the power of two sits in an SGPR the compiler cannot see through
                                uint32_t pw = 1u << sh;
                                asm("" : "+s"(pw));
                                acc += (uint64_t)u[jh] * pw;
#else
                                acc += (uint64_t)u[jh] << sh;
#endif
                                pin(acc);
                            }
                        });
                    } else {
                        constexpr uint32_t dlo = (uint32_t)((unsigned long long)d & HM), dhi = (uint32_t)((unsigned long long)d >> H);
                        static_assert(((unsigned long long)d >> (2 * H)) == 0, "prime limb wider than the radix");
                        static_for<0, M>([&](auto JJ) {
                            constexpr int jh = JJ;
                            if constexpr (2 * l + jh == k && jh < k && dlo != 0) { acc += (uint64_t)u[jh] * dlo; pin(acc); }
                            if constexpr (2 * l + jh + 1 == k && jh < k && dhi != 0) { acc += (uint64_t)u[jh] * dhi; pin(acc); }
                        });
                    }
                }
            });
            if constexpr (k < 4 * N - 2) {
                u[k] = (uint32_t)acc & HM;
                cy = acc >> H;
            } else {
                c[N - 1] = acc;                                               // the last limb is not masked (monty.py:830-838)
            }
        });
        static_for<0, N - 1>([&](auto I) {
            constexpr int i = I;
            c[i] = (spint)u[M + 2 * i] | ((spint)u[M + 2 * i + 1] << H);
        });
    }

    // The trinomial shape with a virtual limb (2^448 - 2^224 - 1: ppw = [-1, 0.., -1 at limb NEG, 0.., +1 at limb N]):
    // the reference carries the negative middle limb with its borrow convention -- column NEG adds Q - v_0, every later
    // column adds the scratch word s = mask - v_{C-NEG} + v_{C-N}, and the last limb adds v_N - 1 (monty.py:597-627,
    // 717-738, 830-838; monty_reduce above).  Each of these is an explicit integer added at the weight of limb column
    // C, so in radix 2^H (R = 2H) it splits exactly: Q - v = (2^H - u_a) + 2^H (2^H - 1 - u_b) and
    // s = (m - u_a + u'_a) + 2^H (m - u_b + u'_b), m = 2^H - 1, all halves non-negative: they go to half columns 2C and
    // 2C + 1 as 32-bit values.  256 multiply-adds into 31 half columns with one carry chain instead of 64 limb products
    // on three accumulators each.
    static constexpr bool MHALF_TRI = FAST && P::MONTGOMERY && P::NDASH == 1 && P::E && RADIX == 56 && N == 8 && P::NEG_LIMB == 4 &&
                                      P::ppw(0) == -1 && P::ppw(4) == -1 && P::ppw(8) == 1 && P::ppw(1) == 0 && P::ppw(2) == 0 &&
                                      P::ppw(3) == 0 && P::ppw(5) == 0 && P::ppw(6) == 0 && P::ppw(7) == 0;
    // a SPLIT that was proven from the sparse term count covers monty_mul / monty_reduce alone (emit.split_is_sparse, ADVICE of round 4:
    // the pairing used to rest on an early return in emit.chain_ok and on MHALF not being selected)
    static_assert(!(SPLIT_SPARSE && (CHAINED || MHALF || MHALF_TRI)), "SPLIT of this prime holds for the sparse columns of monty_mul only");
    template <bool SQR>
    static MA_DEV void monty_mul_half_tri(const spint* a, const spint* b, spint* c) {
        constexpr int H = RADIX / 2, M = 2 * N, NEG = P::NEG_LIMB;
        constexpr uint32_t HM = (1u << H) - 1u;
        uint32_t f[M], g[M], u[4 * N];
        static_for<0, N>([&](auto K) {
            constexpr int k = K;
            f[2 * k] = (uint32_t)a[k] & HM;
            f[2 * k + 1] = (uint32_t)(a[k] >> H);
            if constexpr (!SQR) {
                g[2 * k] = (uint32_t)b[k] & HM;
                g[2 * k + 1] = (uint32_t)(b[k] >> H);
            }
        });
        uint32_t f2[M];
        if constexpr (SQR) static_for<0, M>([&](auto I) { f2[I] = 2u * f[I]; });
        uint64_t cy = 0;
        // limb columns 0 .. 2N-1 = half columns 0 .. 4N-1; digits v_0 .. v_N are half digits u_0 .. u_{2N+1}
        static_for<0, 4 * N>([&](auto KK) {
            constexpr int k = KK, C = k / 2, h = k % 2;
            uint64_t acc = cy;
            constexpr int lo = k < M ? 0 : k - (M - 1), hi = k < M ? k : M - 1;
            if constexpr (lo <= hi) {
                static_for<lo, hi + 1>([&](auto II) {
                    constexpr int i = II, j = k - i;
                    if constexpr (!SQR) {
                        acc += (uint64_t)f[i] * g[j];
                        pin(acc);
                    } else if constexpr (i <= j) {
                        acc += (uint64_t)((i < j) ? f2[i] : f[i]) * f[j];
                        pin(acc);
                    }
                });
            }
            if constexpr (C == NEG) {                                  // Q - v_0
                acc += (uint64_t)(h == 0 ? (HM + 1u) - u[0] : HM - u[1]);
            } else if constexpr (C > NEG) {                            // s = mask - v_{C-NEG} + v_{C-N}
                uint32_t sh = HM;
                if constexpr (C - NEG <= N) sh -= u[2 * (C - NEG) + h];
                if constexpr (C >= N) sh += u[2 * (C - N) + h];
                acc += (uint64_t)sh;
            }
            u[k] = (uint32_t)acc & HM;
            cy = acc >> H;
        });
        const spint vN = (spint)u[2 * N] | ((spint)u[2 * N + 1] << H);
        static_for<0, N - 1>([&](auto I) {
            constexpr int i = I;
            c[i] = (spint)u[2 * (N + 1) + 2 * i] | ((spint)u[2 * (N + 1) + 2 * i + 1] << H);
        });
        c[N - 1] = cy + vN - (spint)1;                                 // monty.py:830-838
    }

    // the same columns on the 64-bit column chain (see Wide::Acc): shifts of a digit become products with a
    // constant power of two (one half of which is zero), one-word terms go to the carry word
    template <int C, class AccT>
    static MA_DEV void monty_reduce_chain(AccT& t, const spint* v, const Opd* V) {
        constexpr int NEG = P::NEG_LIMB;
        constexpr bool scratch = (NEG > 0) && (C > NEG);
        spint s = MASK;
        static_for<1, LMAX + 1>([&](auto L) {
            constexpr int l = L;
            constexpr int j = C - l;
            if constexpr (j >= 0 && j <= JMAX && j < C) {
                constexpr long long d = P::ppw(l);
                if constexpr (d > 1) {
                    t.mac(V[j], W::template prep_const<(unsigned long long)d>());
                } else if constexpr (d == 1) {
                    if constexpr (scratch) s += v[j]; else t.add(v[j]);
                } else if constexpr (d == -1) {
                    if constexpr (scratch) s -= v[j];
                    else t.add(Q - v[j]);
                } else {
                    static_assert(d == 0, "negative prime limbs other than -1 are not supported");
                }
            }
        });
        if constexpr (scratch) t.add(s);
    }
    template <bool SQR>
    static MA_DEV void monty_mul_chain(const spint* a, const spint* b, spint* c) {
        constexpr int NCOL = P::E ? 2 * N : 2 * N - 1;
        typename W::template Acc<RADIX> t;
        spint v[JMAX + 1];
        Opd V[JMAX + 1];
        Opd A[N], B[N];
        static_for<0, N>([&](auto I) { A[I] = W::prep(a[I]); });
        if constexpr (!SQR) static_for<0, N>([&](auto I) { B[I] = W::prep(b[I]); });
        static_for<0, NCOL>([&](auto CC) {
            constexpr int col = CC;
            constexpr int lo = col < N ? 0 : col - (N - 1);
            constexpr int hi = col < N ? col : N - 1;
            if constexpr (lo <= hi) {
                if constexpr (!SQR) {
                    static_for<lo, hi + 1>([&](auto K) {
                        constexpr int k = K;
                        t.mac(A[k], B[col - k]);
                    });
                } else {
                    constexpr int pairs = (hi - lo + 1) / 2;
                    Col cross;
                    static_for<0, pairs>([&](auto J) {
                        constexpr int k = lo + J;
                        cross.mac(A[k], A[col - k]);
                    });
                    if constexpr (pairs > 0) t.add_twice(cross);
                    if constexpr (col % 2 == 0) t.mac(A[col / 2], A[col / 2]);
                }
            }
            monty_reduce_chain<col>(t, v, V);
            if constexpr (col <= JMAX) {
                if constexpr (P::NDASH == 1) {
                    v[col] = t.digit();
                } else {
                    static_assert(P::ppw(0) > 0, "full Montgomery reduction expects a positive low prime limb");
                    v[col] = (t.low() * (spint)P::NDASH) & MASK;
                    if constexpr (P::ppw(0) == 1) t.add(v[col]);
                    else t.mac(W::prep(v[col]), W::prep((spint)P::ppw(0)));
                    (void)t.digit();       // the low Radix bits are zero now
                }
                V[col] = W::prep(v[col]);
            } else {
                c[col - JMAX - 1] = t.digit();
            }
        });
        spint top = t.c;
        if constexpr (P::E) {
            if constexpr (P::NEG_LIMB > 0) top += v[N] - (spint)1;
            else top += v[N];
        } else {
            if constexpr (P::NEG_LIMB > 0) top -= (spint)1;
        }
        c[N - 1] = top;
    }

    // monty.py:876-978: trinomial fold, else Barrett-Dhem
    static MA_DEV void monty_modmli(const spint* a, int b, spint* c) {
        const dpint bw = (dpint)(__int128)b;
        dpint t = 0;
        if constexpr (P::TRIN > 0) {
            static_for<0, N>([&](auto I) {
                t += (dpint)a[I] * bw;
                c[I] = (spint)t & MASK;
                t >>= RADIX;
            });
            spint s = (spint)t;
            if constexpr (P::XCESS > 0) {
                s = (s << P::XCESS) + (c[N - 1] >> (RADIX - P::XCESS));
                c[N - 1] &= ((spint)1 << (RADIX - P::XCESS)) - 1;
            }
            c[0] += s;
            c[P::TRIN] += s;
        } else {
            static_for<0, N - 1>([&](auto I) {
                t += (dpint)a[I] * bw;
                c[I] = (spint)t & MASK;
                t >>= RADIX;
            });
            t += (dpint)a[N - 1] * bw;
            c[N - 1] = (spint)t;
            spint h = (spint)(t >> P::BARRETT_SHIFT);
            spint q = (spint)(((dpint)h * (dpint)(spint)P::BARRETT_R) >> 64);
            bool propc = P::ppw(0) > 0;
            static_for<0, N>([&](auto I) {
                constexpr int i = I;
                constexpr long long d = P::ppw(i);
                if constexpr (i >= 1 && i < N - 1 && d != 0) propc = true;
                if constexpr (d == -1) c[i] += q;
                else if constexpr (d == 1) c[i] -= q;
                else if constexpr (d < -1) {                           // PM form: the low limb -M (monty.py:958-959)
                    static_assert(i == 0 && i < N - 1, "a negative prime limb other than -1 is the low limb of the PM form");
                    dpint w = (dpint)q * (dpint)(spint)(-d);
                    c[i] += (spint)w & MASK;
                    c[i + 1] += (spint)(w >> RADIX);
                } else if constexpr (d > 1 && (d & (d - 1)) == 0) {
                    constexpr int e = __builtin_ctzll((unsigned long long)d);
                    if constexpr (i < N - 1) {
                        dpint w = (dpint)q << e;
                        c[i] -= (spint)w & MASK;
                        c[i + 1] -= (spint)(w >> RADIX);
                    } else {
                        c[i] -= q << e;
                    }
                } else if constexpr (d > 1) {
                    if constexpr (i < N - 1) {
                        dpint w = (dpint)q * (dpint)(spint)d;
                        c[i] -= (spint)w & MASK;
                        c[i + 1] -= (spint)(w >> RADIX);
                    } else {
                        c[i] -= q * (spint)d;
                    }
                } else {
                    static_assert(d == 0, "negative prime limbs other than -1 are not supported");
                }
            });
            if constexpr (P::E) c[N - 1] -= q << RADIX;
            if (propc) (void)prop(c);
        }
    }

    // ================================================================ family dispatch
    static MA_DEV void modmul(const spint* a, const spint* b, spint* c) {
        if constexpr (P::MONTGOMERY) {
            if constexpr (MHALF_TRI) monty_mul_half_tri<false>(a, b, c);
            else if constexpr (MHALF) monty_mul_half<false>(a, b, c);
            else if constexpr (CHAINED) monty_mul_chain<false>(a, b, c);
            else monty_mul<false>(a, b, c);
        } else {
            if constexpr (HALF_OV) pm_mul_half_ov<false>(a, b, c);
            else if constexpr (HALF) pm_modmul_half(a, b, c);
            else if constexpr (CHAINED) pm_modmul_chain(a, b, c);
            else if constexpr (SPLIT4) pm_modmul_split4(a, b, c);
            else pm_modmul(a, b, c);
        }
    }
    static MA_DEV void modsqr(const spint* a, spint* c) {
        if constexpr (P::MONTGOMERY) {
            if constexpr (MHALF_TRI) monty_mul_half_tri<true>(a, a, c);
            else if constexpr (MHALF) monty_mul_half<true>(a, a, c);
            else if constexpr (CHAINED) monty_mul_chain<true>(a, a, c);
            else monty_mul<true>(a, a, c);
        } else {
            if constexpr (HALF_OV) pm_mul_half_ov<true>(a, a, c);
            else if constexpr (HALF) pm_modsqr_half(a, c);
            else if constexpr (CHAINED) pm_modsqr_chain(a, c);
            else if constexpr (SPLIT4) pm_modsqr_split4(a, c);
            else pm_modsqr(a, c);
        }
    }
    static MA_DEV void modmli(const spint* a, int b, spint* c) {
        if constexpr (P::MONTGOMERY) monty_modmli(a, b, c); else pm_modmli(a, b, c);
    }
    // pseudo.py:952-962 (copy) / monty.py:1386-1399 (multiply by R^2 mod p)
    static MA_DEV void nres(const spint* m, spint* n) {
        if constexpr (P::MONTGOMERY) {
            spint r2[N];
            static_for<0, N>([&](auto I) { r2[I] = P::r2(I); });
            modmul(m, r2, n);
        } else {
            modcpy(m, n);
        }
    }
    // pseudo.py:965-976 / monty.py:1402-1416
    static MA_DEV void redc(const spint* n, spint* m) {
        if constexpr (P::MONTGOMERY) {
            spint one[N];
            static_for<0, N>([&](auto I) { one[I] = (I == 0) ? 1 : 0; });
            modmul(n, one, m);
        } else {
            modcpy(n, m);
        }
        (void)modfsb(m);
    }

    // rolled on purpose: the chains call this with constant counts up to a few hundred, and a fully unrolled
    // progenitor is tens of thousands of instructions per prime
    static MA_DEV void modnsqr(spint* a, int n) {
#pragma unroll 1
        for (int i = 0; i < n; i++) modsqr(a, a);
    }

    // progenitor x^PE (pseudo.py:758-785).  The reference takes its addition chain from the external
    // `addchain` tool; ours comes from the driver (P::modpro_chain).  Limbs differ, values do not.
    // The chain is ~250 squarings + a dozen multiplications.  It is built from TWO out-of-line primitives -- a rolled run
    // of squarings and one multiplication (chain_nsqr, chain_mul: real device functions, one copy each per field and
    // policy in a translation unit) -- that every step of the chain calls.  Round 2 inlined the whole chain into one
    // non-inlined function: a dozen copies of both bodies in a function without a register bound, which the scheduler
    // filled to 228-248 VGPRs (1-2 waves per SIMD for every kernel that calls it) unless the multiply-add chains were
    // left unpinned -- and unpinned, LLVM's reassociation turns a 110-instruction squaring into 264 instructions
    // (PMC: 67 000 VALU instructions per X25519 modpro instead of 31 000).  Out of line, each primitive keeps the pinned
    // product (the same code as the streaming kernels) in a small function of its own; the operands cross the call in
    // private memory, 10 / 30 dword accesses per call against thousands of cycles of arithmetic.
    static __device__ __attribute__((noinline)) void chain_nsqr(spint* a, int n) {
        spint x[N];
        modcpy(a, x);
#pragma unroll 1
        for (int i = 0; i < n; i++) modsqr(x, x);
        modcpy(x, a);
    }
    static __device__ __attribute__((noinline)) void chain_mul(const spint* a, const spint* b, spint* c) {
        spint x[N], y[N], z[N];
        modcpy(a, x);
        modcpy(b, y);
        modmul(x, y, z);
        modcpy(z, c);
    }
    struct ChainOps {           // what P::modpro_chain<F> calls
        static MA_DEV void modcpy(const spint* a, spint* c) { Field::modcpy(a, c); }
        static MA_DEV void modnsqr(spint* a, int n) { Field::chain_nsqr(a, n); }
        static MA_DEV void modsqr(const spint* a, spint* c) { if (a != c) Field::modcpy(a, c); Field::chain_nsqr(c, 1); }
        static MA_DEV void modmul(const spint* a, const spint* b, spint* c) { Field::chain_mul(a, b, c); }
    };
#ifdef MA_MODPRO_ROUND2         // experiment switch: the round-2 form (whole chain inlined into one non-inlined function, unpinned)
    static __device__ __attribute__((noinline)) void modpro(const spint* w, spint* z) {
        spint x[N], r[N];
        modcpy(w, x);
        P::template modpro_chain<Field<P, FAST_, false>>(x, r);
        modcpy(r, z);
    }
#else
    static MA_DEV void modpro(const spint* w, spint* z) {
        spint x[N], r[N];
        modcpy(w, x);
        P::template modpro_chain<ChainOps>(x, r);
        modcpy(r, z);
    }
#endif

    // The products AROUND the progenitor in modinv / modqr / modsqrt go through the out-of-line primitives of the chain too (round 5):
    // inlined, two modinv of a 13-limb field in one kernel (k_time's third leg) came to 512 registers and 34 432 spilled ones, modqr to
    // 244, the Tonelli-Shanks tail of modsqrt over 2^255-19 to 26 spilled at its three-wave budget and 324 registers inside ecn set --
    // for a handful of products next to the ~250-750 of the chain, which has been out of line since round 3.
    static constexpr bool TAIL_OOL = true;
    static MA_DEV void tail_mul(const spint* a, const spint* b, spint* c) { if constexpr (TAIL_OOL) chain_mul(a, b, c); else modmul(a, b, c); }
    static MA_DEV void tail_sqr(const spint* a, spint* c) {
        if constexpr (TAIL_OOL) { if (a != c) modcpy(a, c); chain_nsqr(c, 1); } else modsqr(a, c);
    }
    static MA_DEV void tail_nsqr(spint* a, int n) { if constexpr (TAIL_OOL) chain_nsqr(a, n); else modnsqr(a, n); }

    // pseudo.py:788-812
    static MA_DEV void modinv(const spint* x, const spint* h, spint* z) {
        spint s[N], t[N];
        if (h == nullptr) modpro(x, t); else modcpy(h, t);
        modcpy(x, s);
        for (int i = 0; i < P::PM1D2 - 1; i++) {
            tail_sqr(s, s);
            tail_mul(s, x, s);
        }
        tail_nsqr(t, P::PM1D2 + 1);
        tail_mul(s, t, z);
    }

    // quadratic-residue test (pseudo.py:815-831)
    static MA_DEV int modqr(const spint* h, const spint* x) {
        spint r[N];
        if (h == nullptr) {
            modpro(x, r);
            tail_sqr(r, r);
        } else {
            tail_sqr(h, r);
        }
        tail_mul(r, x, r);
        if constexpr (P::PM1D2 > 1) tail_nsqr(r, P::PM1D2 - 1);
        return modis1(r) | modis0(x);
    }
    // square root: Tonelli-Shanks on the progenitor (pseudo.py:834-874); the data-dependent choice is
    // a lane-predicated modcmv, never a branch
    static MA_DEV void modsqrt(const spint* x, const spint* h, spint* r) {
        spint s[N], y[N];
        if (h == nullptr) modpro(x, y); else modcpy(h, y);
        tail_mul(y, x, s);
        if constexpr (P::PM1D2 > 1) {
            spint t[N], b[N], v[N], z[N];
            static_for<0, N>([&](auto I) { z[I] = P::roi(I); });
            tail_mul(s, y, t);
            nres(z, z);
            for (int k = P::PM1D2; k > 1; k--) {
                modcpy(t, b);
                tail_nsqr(b, k - 2);
                int d = 1 - modis1(b);
                tail_mul(s, z, v);
                modcmv(d, v, s);
                tail_sqr(z, z);
                tail_mul(t, z, v);
                modcmv(d, v, t);
            }
        }
        modcpy(s, r);
    }

    // pseudo.py:877-906
    static MA_DEV int modis1(const spint* a) {
        spint c[N], d = 0;
        redc(a, c);
        static_for<1, N>([&](auto I) { d |= c[I]; });
        return (int)((spint)1 & ((d - (spint)1) >> RADIX) & (((c[0] ^ (spint)1) - (spint)1) >> RADIX));
    }
    static MA_DEV int modis0(const spint* a) {
        spint c[N], d = 0;
        redc(a, c);
        static_for<0, N>([&](auto I) { d |= c[I]; });
        return (int)((spint)1 & ((d - (spint)1) >> RADIX));
    }
    static MA_DEV void modzer(spint* a) { static_for<0, N>([&](auto I) { a[I] = 0; }); }
    static MA_DEV void modint(int x, spint* a) {
        a[0] = (spint)x;
        static_for<1, N>([&](auto I) { a[I] = 0; });
        nres(a, a);
    }
    static MA_DEV void modone(spint* a) { modint(1, a); }

    // Constant-time conditional move / swap by lane predication (v_cndmask): no data-dependent
    // branch or address.  Equal to the reference's PSCR arithmetic form for b in {0,1}
    // (pseudo.py:979-1048; see oracle/field_common.inc for the algebra).
    static MA_DEV void modcmv(int b, const spint* g, spint* f) {
        const bool take = (b & 1) != 0;
        // both values are loaded before the choice: `take ? g[I] : f[I]` is an lvalue conditional, i.e. a choice
        // of ADDRESS, which sends private arrays to scratch behind flat pointers
        static_for<0, N>([&](auto I) {
            const spint x = g[I], y = f[I];
            f[I] = take ? x : y;
        });
    }
    static MA_DEV void modcsw(int b, spint* g, spint* f) {
        const bool take = (b & 1) != 0;
        static_for<0, N>([&](auto I) {
            spint x = g[I], y = f[I];
            g[I] = take ? y : x;
            f[I] = take ? x : y;
        });
    }

    // pseudo.py:1052-1081
    static MA_DEV void modshl(unsigned n, spint* a) {
        a[N - 1] = (a[N - 1] << n) + (a[N - 2] >> (RADIX - n));
        static_for<0, N - 2>([&](auto J) {
            constexpr int i = N - 2 - J;
            a[i] = ((a[i] << n) & MASK) + (a[i - 1] >> (RADIX - n));
        });
        a[0] = (a[0] << n) & MASK;
    }
    static MA_DEV int modshr(unsigned n, spint* a) {
        spint r = a[0] & (((spint)1 << n) - (spint)1);
        static_for<0, N - 1>([&](auto I) {
            constexpr int i = I;
            a[i] = (a[i] >> n) + ((a[i + 1] << (RADIX - n)) & MASK);
        });
        a[N - 1] = a[N - 1] >> n;
        return (int)r;
    }
    // pseudo.py:1084-1100
    static MA_DEV void modhaf(spint* n) {
        spint t[N];
        (void)prop(n);
        modcpy(n, t);
        int lsb = modshr(1, t);
        addp<1>(n, ~(spint)0);
        (void)prop(n);
        (void)modshr(1, n);
        modcmv(1 - lsb, t, n);
    }
    // pseudo.py:1102-1112, monty.py:1578-1593
    static MA_DEV void mod2r(unsigned r, spint* a) {
        modzer(a);
        if (r >= (unsigned)P::NBYTES * 8) return;
        unsigned n = r / RADIX, m = r % RADIX;
        static_for<0, N>([&](auto I) { a[I] = ((unsigned)I == n) ? ((spint)1 << m) : 0; });
        if constexpr (P::MONTGOMERY) nres(a, a);
    }
    // pseudo.py:1149-1174
    static MA_DEV int modsign(const spint* a) {
        spint c[N];
        redc(a, c);
        return (int)(c[0] % 2);
    }
    static MA_DEV int modcmp(const spint* a, const spint* b) {
        spint c[N], d[N];
        int eq = 1;
        redc(a, c);
        redc(b, d);
        static_for<0, N>([&](auto I) { eq &= (int)((((c[I] ^ d[I]) - 1) >> RADIX) & 1); });
        return eq;
    }

    // Byte import/export.  The reference shifts in 8 bits at a time (pseudo.py:1115-1146); the limbs
    // it ends with are a pure function of the integer the bytes spell: limb i = bits
    // [RADIX*i, RADIX*(i+1)), the top limb taking every remaining bit, then modfsb + nres as there.
    // Here the integer arrives / leaves as NW little-endian 64-bit words held in registers; kernels
    // turn big-endian (modimp/modexp) or little-endian (rfc7748) byte records into such words.
    static constexpr int NW = (P::NBYTES + 7) / 8;   // a short top word is zero-extended

    static MA_DEV void limbs_from_words(const spint* w, spint* a) {
        static_for<0, N>([&](auto I) {
            constexpr int i = I;
            constexpr int o = RADIX * i, wi = o / 64, sh = o % 64;
            spint val = 0;
            if constexpr (wi < NW) {
                val = w[wi] >> sh;
                if constexpr (sh + RADIX > 64 && wi + 1 < NW) val |= w[wi + 1] << (64 - sh);
            }
            if constexpr (i < N - 1) val &= MASK;
            a[i] = val;
        });
    }
    // limbs must be canonical (< 2^RADIX each), as redc leaves them
    static MA_DEV void words_from_limbs(const spint* c, spint* w) {
        static_for<0, NW>([&](auto K) { w[K] = 0; });
        static_for<0, N>([&](auto I) {
            constexpr int i = I;
            constexpr int o = RADIX * i, wi = o / 64, sh = o % 64;
            if constexpr (wi < NW) {
                w[wi] |= c[i] << sh;
                if constexpr (sh + RADIX > 64 && wi + 1 < NW) w[wi + 1] |= c[i] >> (64 - sh);
            }
        });
    }
    // returns 1 if the value was < p (pseudo.py:1130-1146)
    static MA_DEV int modimp_words(const spint* w, spint* a) {
        limbs_from_words(w, a);
        int res = (int)modfsb(a);
        nres(a, a);
        return res;
    }
    // pseudo.py:1115-1127
    static MA_DEV void modexp_words(const spint* a, spint* w) {
        spint c[N];
        redc(a, c);
        words_from_limbs(c, w);
    }
};

}  // namespace ma
