// modarith_amd/csrc/kernels.h -- batched element-wise field kernels for gfx950.
//
// HBM layout: limb-interleaved SoA, buf[limb * ld + j] (u64), the n-lane generalisation of the
// reference's SIMD "batched form" (simd/pseudo_simd.py: a[i] holds limb i of every lane).  One field
// element per lane; lane j of a wave touches consecutive u64 of each limb row, so every limb load
// is one fully coalesced 512-byte (EPT=1) or 1-KiB (EPT=2, 16 B per lane) wave access.  All limbs of
// the operands are loaded before any arithmetic (independent loads in flight), results are stored
// limb row by limb row.  The kernels are HBM-bound streaming kernels: grid-stride over at most
// MA_MAX_BLOCKS workgroups of 256 threads.  No LDS is needed for per-lane data; the shared-multiplicand
// variant broadcasts its common operand from kernel arguments (SGPRs), which is cheaper than LDS.
#pragma once
#include "field.h"

namespace ma {

constexpr int BLOCK = 256;

// Batches are streamed once (640 MiB+ per array, far beyond L2 / Infinity Cache): loads and stores carry
// the non-temporal hint so they do not displace each other in the caches (measured +3..5 % on the
// 3-stream pattern, profiles/history/r01_membench.log).
typedef spint spint2 __attribute__((ext_vector_type(2)));
#ifndef MA_NONTEMPORAL
#define MA_NONTEMPORAL 1
#endif
template <class T> __device__ __forceinline__ T ld_stream(const T* p) {
#if MA_NONTEMPORAL
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
template <class T> __device__ __forceinline__ void st_stream(T* p, T v) {
#if MA_NONTEMPORAL
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// Limb stride descriptor of a batch.  FLAT (ld >= n, the n-lane form of the reference's SIMD "batched form"): limb i of element
// j at buf[i * ld + j].  TILED (ld = 2^s < n): the batch is a sequence of tiles of ld elements, each tile limb-interleaved with
// stride ld, tiles N * ld words apart -- buf[((j >> s) * N + i) * ld + (j & (ld - 1))].  One formula serves both: the flat case
// carries s = 63, so that j >> s = 0 and j & (2^s - 1) = j.  Why tiles: with rows of 2^12 elements (32 KiB) the N limb rows a
// workgroup streams lie within one contiguous 160 KiB (5 limbs) / 256 KiB (8 limbs) stretch instead of N stretches 128 MiB
// apart, and the streaming rate no longer depends on where the driver placed the arrays (profiles/history/r03_tiled_exp_*).
struct Ld {
    size_t ld;
    unsigned s;
    __host__ __device__ Ld(size_t ld_ = 0) : ld(ld_), s(63) {}
    __host__ __device__ Ld(size_t ld_, unsigned s_) : ld(ld_), s(s_) {}
    template <int N>
    __host__ __device__ __forceinline__ size_t off(size_t j) const { return (((j >> s) * (size_t)N) << s) + (j & ((((size_t)1) << s) - 1)); }
};

// element index handled by (thread t, slot e): j = EPT*t + e  -> contiguous EPT*8 bytes per lane
template <class P, int EPT>
__device__ __forceinline__ void load_soa(const spint* base, Ld L, size_t t, spint (*x)[P::N]) {
    const spint* p = base + L.template off<P::N>((size_t)EPT * t);
    if constexpr (EPT == 1) {
        static_for<0, P::N>([&](auto I) { x[0][I] = ld_stream(p + (size_t)I * L.ld); });
    } else {
        static_for<0, P::N>([&](auto I) {
            spint2 v = ld_stream(reinterpret_cast<const spint2*>(p + (size_t)I * L.ld));
            x[0][I] = v.x;
            x[1][I] = v.y;
        });
    }
}
template <class P, int EPT>
__device__ __forceinline__ void store_soa(spint* base, Ld L, size_t t, spint (*x)[P::N]) {
    spint* p = base + L.template off<P::N>((size_t)EPT * t);
    if constexpr (EPT == 1) {
        static_for<0, P::N>([&](auto I) { st_stream(p + (size_t)I * L.ld, x[0][I]); });
    } else {
        static_for<0, P::N>([&](auto I) {
            spint2 v;
            v.x = x[0][I];
            v.y = x[1][I];
            st_stream(reinterpret_cast<spint2*>(p + (size_t)I * L.ld), v);
        });
    }
}

// ---- operation functors: apply() works on register-resident elements
template <class P, bool FAST = false> struct OpMul { static MA_DEV void apply(const spint* a, const spint* b, spint* c) { Field<P, FAST>::modmul(a, b, c); } };
template <class P> struct OpAdd { static MA_DEV void apply(const spint* a, const spint* b, spint* c) { Field<P>::modadd(a, b, c); } };
template <class P> struct OpSub { static MA_DEV void apply(const spint* a, const spint* b, spint* c) { Field<P>::modsub(a, b, c); } };
template <class P> struct OpAddLazy { static MA_DEV void apply(const spint* a, const spint* b, spint* c) { Field<P>::modadd_lazy(a, b, c); } };
template <class P> struct OpSubLazy { static MA_DEV void apply(const spint* a, const spint* b, spint* c) { Field<P>::modsub_lazy(a, b, c); } };
// modmul / modsqr with a wave-uniform choice of product policy: the split products (Field<P,true>, fewer VALU
// instructions) when every active lane's operands are inside their limb contract (< 2^(Radix+2), field.h), the exact
// 128-bit products otherwise -- identical results either way, for every input
// (radix 62 and up: 2^(Radix+2) does not fit a word, every 64-bit limb is below it -- the predicate is vacuously true,
// and a shift by 64 would be undefined)
template <class P> MA_DEV bool in_split_contract(const spint* a) {
    if constexpr (P::RADIX + 2 >= 64) {
        return true;
    } else {
        spint m = 0;
        static_for<0, P::N>([&](auto I) { m |= a[I]; });
        return (m >> (P::RADIX + 2)) == 0;
    }
}
// the budget modlimbs / Curve.limbs_ok report: 2^(Radix+2) as above, except for the 5 x 52-bit pseudo-Mersenne primes whose curve
// kernels run the folded half-limb products (field.h FOLD52, 2^256-189): there the reference itself wraps mm * a_k at 64
// bits, and the two product forms agree only for limbs up to (2^64-1)/mm (2^52.4) -- which every field-function output keeps
template <class P> MA_DEV bool in_limb_budget(const spint* a) {
    constexpr bool fold52 = !P::MONTGOMERY && P::EPM && !P::OVERFLOW && P::RADIX == 52 && P::N == 5 && P::MM < (1ull << 16) && P::SPLIT == 0;
    if constexpr (fold52) {
        bool ok = true;
        static_for<0, P::N>([&](auto I) { ok = ok & (a[I] <= ~(spint)0 / (spint)P::MM); });      // (&: no short-circuit branch per limb)
        return ok;
    } else {
        return in_split_contract<P>(a);
    }
}
template <class P> struct OpMulAuto {
    static MA_DEV void apply(const spint* a, const spint* b, spint* c) {
        if constexpr (P::SPLIT > 0) {
            if (__all(in_split_contract<P>(a) && in_split_contract<P>(b))) { Field<P, true>::modmul(a, b, c); return; }
        }
        Field<P, false>::modmul(a, b, c);
    }
};
template <class P> struct OpSqrAuto {
    static MA_DEV void apply(const spint* a, spint* c) {
        if constexpr (P::SPLIT > 0) {
            if (__all(in_split_contract<P>(a))) { Field<P, true>::modsqr(a, c); return; }
        }
        Field<P, false>::modsqr(a, c);
    }
};
template <class P> struct OpNresAuto {
    static MA_DEV void apply(const spint* a, spint* c) {
        if constexpr (P::SPLIT > 0 && P::MONTGOMERY) {
            if (__all(in_split_contract<P>(a))) { Field<P, true>::nres(a, c); return; }
        }
        Field<P, false>::nres(a, c);
    }
};
template <class P> struct OpRedcAuto {
    static MA_DEV void apply(const spint* a, spint* c) {
        if constexpr (P::SPLIT > 0 && P::MONTGOMERY) {
            if (__all(in_split_contract<P>(a))) { Field<P, true>::redc(a, c); return; }
        }
        Field<P, false>::redc(a, c);
    }
};
// The 8-byte-per-lane kernels (EPT = 1: unaligned buffers, odd limb strides, the last element of an odd batch) run the exact
// products instead of the per-wave choice: compiled with EPT = 1 the two-path functors were allocated 217 (5 limbs) to 512
// (8 limbs: 256 VGPRs + 256 AGPRs, a v_accvgpr copy around every multiply-add) registers, and unaligned X448 batches
// streamed at 2.0 TB/s.  The exact path is right for every input, needs 66-100 VGPRs, and at 8 bytes per lane the kernel
// is not faster than its arithmetic anyway.
template <class Op> struct ScalarOp { using type = Op; };
template <class P> struct ScalarOp<OpMulAuto<P>> { using type = OpMul<P, false>; };
template <class P, bool FAST = false> struct OpSqr { static MA_DEV void apply(const spint* a, spint* c) { Field<P, FAST>::modsqr(a, c); } };
template <class P> struct ScalarOp<OpSqrAuto<P>> { using type = OpSqr<P, false>; };
template <class P> struct OpNeg { static MA_DEV void apply(const spint* a, spint* c) { Field<P>::modneg(a, c); } };
template <class P> struct OpNegLazy { static MA_DEV void apply(const spint* a, spint* c) { Field<P>::modneg_lazy(a, c); } };
template <class P, bool FAST = false> struct OpNres { static MA_DEV void apply(const spint* a, spint* c) { Field<P, FAST>::nres(a, c); } };
template <class P, bool FAST = false> struct OpRedc { static MA_DEV void apply(const spint* a, spint* c) { Field<P, FAST>::redc(a, c); } };
template <class P> struct ScalarOp<OpNresAuto<P>> { using type = OpNres<P, false>; };
template <class P> struct ScalarOp<OpRedcAuto<P>> { using type = OpRedc<P, false>; };
template <class P> struct OpCpy { static MA_DEV void apply(const spint* a, spint* c) { Field<P>::modcpy(a, c); } };
// modinv of the batched API returns the inverse in NORMALISED form nres(redc(1/a)): limbs that are a function of the value alone
// (canonical limbs for the pseudo-Mersenne fields, the Montgomery form of the canonical value otherwise).  The reference's
// modinv leaves whatever its addition chain leaves -- a chain this library does not share (the reference's comes from an
// external tool), so its limbs were never comparable before redc -- and normalising makes the per-element kernel, the
// progenitor form and the simultaneous-inversion kernel (k_inv_simul) return the same words for the same value.
template <class F> MA_DEV void inv_normalise(spint* z) {
    spint t[F::N];
    F::redc(z, t);
    F::nres(t, z);
}
template <class P, bool FAST = false> struct OpInv {
    static MA_DEV void apply(const spint* a, spint* c) {
        Field<P, FAST>::modinv(a, nullptr, c);
        inv_normalise<Field<P, FAST>>(c);          // (1/a is a product of field-function outputs: inside the limb contract)
    }
};
template <class P, bool FAST = false> struct OpSqrt { static MA_DEV void apply(const spint* a, spint* c) { Field<P, FAST>::modsqrt(a, nullptr, c); } };
template <class P, bool FAST = false> struct OpPro { static MA_DEV void apply(const spint* a, spint* c) { Field<P, FAST>::modpro(a, c); } };
// the long chains (x^PE and what hangs on it: ~250-450 squarings + multiplications per element) with the same wave-uniform
// choice as OpMulAuto: split / half-limb products when every lane's input is inside the limb contract (the chain then stays
// inside it by closure), exact ones otherwise -- the same limbs either way
template <class P, template <class, bool> class Op> struct OpAutoUnary {
    static MA_DEV void apply(const spint* a, spint* c) {
        if constexpr (P::SPLIT > 0) {
            if (__all(in_split_contract<P>(a))) { Op<P, true>::apply(a, c); return; }
        }
        Op<P, false>::apply(a, c);
    }
};

// c[j] = op(a[j], b[j])
template <class P, class Op, int EPT>
__global__ __launch_bounds__(BLOCK) void k_binary(const spint* a, const spint* b,
                                                  spint* c, size_t nthreads, Ld lda, Ld ldb, Ld ldc) {
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < nthreads; t += (size_t)gridDim.x * BLOCK) {
        spint x[EPT][P::N], y[EPT][P::N], z[EPT][P::N];
        load_soa<P, EPT>(a, lda, t, x);
        load_soa<P, EPT>(b, ldb, t, y);
        // (written out, not `#pragma unroll`: for the 8-limb two-path functors the optimizer declined to unroll the loop --
        // "loop not unrolled" for ED448Q / SIDH434 -- which leaves x[e] indexed at run time, i.e. in scratch)
        static_assert(EPT == 1 || EPT == 2, "one or two elements per lane");
        Op::apply(x[0], y[0], z[0]);
        if constexpr (EPT == 2) Op::apply(x[1], y[1], z[1]);
        store_soa<P, EPT>(c, ldc, t, z);
    }
}

// c[j] = op(a[j]) for the long chains (modinv, modsqrt, modpro): same body as k_unary below, but compiled for at least
// three waves per SIMD on the small fields -- with only __launch_bounds__(256) the register allocator may take 512
// VGPRs, and it does (400 for the pinned half-limb products of modinv): one wave per SIMD on a latency-bound chain
template <class P, class Op>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(P::N <= 5 ? 3 : 1)))
void k_unary_heavy(const spint* a, spint* c, size_t nthreads, Ld lda, Ld ldc) {
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < nthreads; t += (size_t)gridDim.x * BLOCK) {
        spint x[1][P::N], z[1][P::N];
        load_soa<P, 1>(a, lda, t, x);
        Op::apply(x[0], z[0]);
        store_soa<P, 1>(c, ldc, t, z);
    }
}

// c[j] = op(a[j])
template <class P, class Op, int EPT>
__global__ __launch_bounds__(BLOCK) void k_unary(const spint* a, spint* c, size_t nthreads,
                                                 Ld lda, Ld ldc) {
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < nthreads; t += (size_t)gridDim.x * BLOCK) {
        spint x[EPT][P::N], z[EPT][P::N];
        load_soa<P, EPT>(a, lda, t, x);
        static_assert(EPT == 1 || EPT == 2, "one or two elements per lane");
        Op::apply(x[0], z[0]);
        if constexpr (EPT == 2) Op::apply(x[1], z[1]);
        store_soa<P, EPT>(c, ldc, t, z);
    }
}

// shared multiplicand: c[j] = a[j] * b0, b0 passed by value (lands in SGPRs, broadcast to all lanes).
// AUTO: the per-wave product policy of OpMulAuto with the vote on a[] only -- the host has already checked b0 against
// the limb contract (capi_prime.inc modmuls), and everything the split / half-limb products derive from b0 alone (its halves,
// mm * half, the prepared Opd pairs) is wave-uniform: the compiler keeps it in SGPRs, computed once per kernel in
// scalar ALU, and every multiply-add reads ONE scalar and one vector source (the VOP3 constant-bus limit), so no
// v_mov / s_nop copies surround the products as they do around the exact 64 x 64 -> 128 ones.  Same limbs either way.
template <class P> struct Elem { spint l[P::N]; };
template <class P, bool AUTO>
MA_DEV void mul_shared_one(const spint* x, const spint* b, spint* z) {
    if constexpr (AUTO && P::SPLIT > 0) {
        if (__all(in_split_contract<P>(x))) { Field<P, true>::modmul(x, b, z); return; }
    }
    Field<P, false>::modmul(x, b, z);
}
template <class P, int EPT, bool AUTO>
__global__ __launch_bounds__(BLOCK) void k_mul_shared(const spint* a, Elem<P> b0, spint* c,
                                                      size_t nthreads, Ld lda, Ld ldc) {
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < nthreads; t += (size_t)gridDim.x * BLOCK) {
        spint x[EPT][P::N], z[EPT][P::N];
        load_soa<P, EPT>(a, lda, t, x);
        // one element at a time, each with its own vote (as OpMulAuto): voting once for both elements lets the scheduler
        // interleave two products and takes 136 VGPRs instead of 64
        static_assert(EPT == 1 || EPT == 2, "one or two elements per lane");
        mul_shared_one<P, AUTO>(x[0], b0.l, z[0]);
        if constexpr (EPT == 2) mul_shared_one<P, AUTO>(x[1], b0.l, z[1]);
        store_soa<P, EPT>(c, ldc, t, z);
    }
}

// c[j] = a[j] * b (small integer)
template <class P, int EPT>
__global__ __launch_bounds__(BLOCK) void k_mli(const spint* a, int b, spint* c, size_t nthreads,
                                               Ld lda, Ld ldc) {
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < nthreads; t += (size_t)gridDim.x * BLOCK) {
        spint x[EPT][P::N], z[EPT][P::N];
        load_soa<P, EPT>(a, lda, t, x);
        static_assert(EPT == 1 || EPT == 2, "one or two elements per lane");
        Field<P>::modmli(x[0], b, z[0]);
        if constexpr (EPT == 2) Field<P>::modmli(x[1], b, z[1]);
        store_soa<P, EPT>(c, ldc, t, z);
    }
}

// a[j] = a[j]^(2^k)
template <class P>
__global__ __launch_bounds__(BLOCK) void k_nsqr(spint* a, int k, size_t n, Ld ld) {
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (size_t)gridDim.x * BLOCK) {
        spint x[1][P::N];
        load_soa<P, 1>(a, ld, t, x);
        bool fast = false;
        if constexpr (P::SPLIT > 0) fast = __all(in_split_contract<P>(x[0]));
        if (fast) Field<P, true>::modnsqr(x[0], k); else Field<P, false>::modnsqr(x[0], k);
        store_soa<P, 1>(a, ld, t, x);
    }
}

// z[j] = 1/x[j] with caller-supplied progenitor h[j] (modinv(x,h,z), pseudo.py:788-812)
template <class P>
__global__ __launch_bounds__(BLOCK) void k_inv_h(const spint* xs, const spint* hs,
                                                 spint* zs, size_t n, Ld ldx, Ld ldh, Ld ldz) {
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (size_t)gridDim.x * BLOCK) {
        spint x[1][P::N], h[1][P::N], z[1][P::N];
        load_soa<P, 1>(xs, ldx, t, x);
        load_soa<P, 1>(hs, ldh, t, h);
        bool fast = false;
        if constexpr (P::SPLIT > 0) fast = __all(in_split_contract<P>(x[0]) && in_split_contract<P>(h[0]));
        if (fast) { Field<P, true>::modinv(x[0], h[0], z[0]); inv_normalise<Field<P, true>>(z[0]); }
        else { Field<P, false>::modinv(x[0], h[0], z[0]); inv_normalise<Field<P, false>>(z[0]); }
        store_soa<P, 1>(zs, ldz, t, z);
    }
}

// z[j] = 1/x[j] for a whole batch with ONE inversion per `rounds` elements (Montgomery's simultaneous inversion).  modinv is
// ~265 field multiplications (38 700 VALU instructions for 2^255-19); a batch does not owe one to every element.  Lane j of L
// takes the elements {r * L + j : r < rounds} (every access of a wave is one coalesced row): forward, it multiplies them up,
// leaving the prefix products c_r in cs (the output buffer itself, or scratch when the output aliases the input);
// then it inverts the last product; backward, 1/x_r = inv * c_{r-1} and inv *= x_r.
// No element may spoil the result of another, whatever its limbs.  Two kinds are kept out of the shared product by lane
// predication (c_r = c_{r-1}), their verdicts travelling to the backward pass in bits 63 / 62 of the stored prefix's top limb
// (free: prefixes are inside the limb contract and the path is taken for radix <= 60 only):
//   * zero values -- tested on the PRODUCT c_{r-1} * x_r, a field-function output, where modis0 is exact whatever the
//     representation of x_r (0, p, even 2p); their output is zero, as modinv(0) = 0 in the reference (pseudo.py:788-812);
//   * elements with limbs outside the contract (fabricated; the reference's behaviour on them is its 64-bit wrap-around):
//     they get an inversion of their own on the exact products in the backward pass -- exactly what the per-element
//     kernel does for a wave that holds one -- paid by that wave only.
// Everything that enters a product is therefore inside the contract: all products run on the split forms.  Outputs in
// normalised form (inv_normalise): the same words as the per-element kernel for every input.
template <class P>
struct InvSimul {
    using F0 = Field<P, false>;
    using F1 = Field<P, (P::SPLIT > 0)>;
    static constexpr spint ZERO = (spint)1 << 63, OOC = (spint)1 << 62;
    static MA_DEV void load(const spint* xs, Ld ldx, size_t e, spint* x) {
        spint t[1][P::N];
        load_soa<P, 1>(xs, ldx, e, t);
        static_for<0, P::N>([&](auto I) { x[I] = t[0][I]; });
    }
    static MA_DEV void run(const spint* xs, spint* zs, spint* cs, size_t n, size_t L, int rounds, Ld ldx, Ld ldz, Ld ldc, size_t j) {
        spint c[P::N], x[P::N], t[1][P::N];
        F1::modone(c);
#pragma unroll 1
        for (int r = 0; r < rounds; r++) {
            const size_t e = (size_t)r * L + j;
            if (e >= n) break;                                  // (e grows with r)
            load(xs, ldx, e, x);
            const bool ooc = !in_split_contract<P>(x);
            F1::modmul(c, x, t[0]);                             // (discarded for an out-of-contract x)
            const bool zero = !ooc && F1::modis0(t[0]) != 0;
            const bool skip = ooc || zero;
            static_for<0, P::N>([&](auto I) { c[I] = skip ? c[I] : t[0][I]; });
            static_for<0, P::N>([&](auto I) { t[0][I] = c[I]; });
            t[0][P::N - 1] |= (zero ? ZERO : (spint)0) | (ooc ? OOC : (spint)0);
            store_soa<P, 1>(cs, ldc, e, t);
        }
        spint inv[P::N];
        F1::modinv(c, nullptr, inv);
#pragma unroll 1
        for (int r = rounds - 1; r >= 0; r--) {
            const size_t e = (size_t)r * L + j;
            if (e >= n) continue;
            load(xs, ldx, e, x);
            const spint flags = cs[ldc.template off<P::N>(e) + (size_t)(P::N - 1) * ldc.ld];
            const bool zero = (flags & ZERO) != 0, ooc = (flags & OOC) != 0;
            spint zi[P::N];
            if (r > 0) {
                load_soa<P, 1>(cs, ldc, e - L, t);
                t[0][P::N - 1] &= ~(ZERO | OOC);
                F1::modmul(inv, t[0], zi);                      // inv * c_{r-1}
                F1::modmul(inv, x, t[0]);
                static_for<0, P::N>([&](auto I) { inv[I] = (zero || ooc) ? inv[I] : t[0][I]; });
            } else {
                static_for<0, P::N>([&](auto I) { zi[I] = inv[I]; });
            }
            inv_normalise<F1>(zi);
            static_for<0, P::N>([&](auto I) { zi[I] = zero ? (spint)0 : zi[I]; });
            if (__any(ooc)) {                                   // fabricated limbs somewhere in this wave: their own inversion, exact products
                spint w[P::N];
                F0::modinv(x, nullptr, w);
                inv_normalise<F0>(w);
                static_for<0, P::N>([&](auto I) { zi[I] = ooc ? w[I] : zi[I]; });
            }
            static_for<0, P::N>([&](auto I) { t[0][I] = zi[I]; });
            store_soa<P, 1>(zs, ldz, e, t);
        }
    }
};
template <class P>
__global__ __launch_bounds__(BLOCK) void k_inv_simul(const spint* xs, spint* zs, spint* cs, size_t n, size_t L, int rounds, Ld ldx, Ld ldz, Ld ldc) {
    const size_t j = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j < L) InvSimul<P>::run(xs, zs, cs, n, L, rounds, ldx, ldz, ldc, j);
}

// r[j] = sqrt(x[j]) / qr(x[j]) with caller-supplied progenitors h[j] (pseudo.py:815-874)
template <class P, bool QR>
__global__ __launch_bounds__(BLOCK) void k_sqrt_h(const spint* xs, const spint* hs, spint* rs, int* out, size_t n, Ld ld) {
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (size_t)gridDim.x * BLOCK) {
        spint x[1][P::N], h[1][P::N], r[1][P::N];
        load_soa<P, 1>(xs, ld, t, x);
        load_soa<P, 1>(hs, ld, t, h);
        bool fast = false;
        if constexpr (P::SPLIT > 0) fast = __all(in_split_contract<P>(x[0]) && in_split_contract<P>(h[0]));
        if constexpr (QR) {
            out[t] = fast ? Field<P, true>::modqr(h[0], x[0]) : Field<P, false>::modqr(h[0], x[0]);
        } else {
            if (fast) Field<P, true>::modsqrt(x[0], h[0], r[0]); else Field<P, false>::modsqrt(x[0], h[0], r[0]);
            store_soa<P, 1>(rs, ld, t, r);
        }
    }
}

// constant-time conditional swap / move with a per-element selector d[j] in {0,1}
// (simd/pseudo_simd.py:1121-1162: selector widened to one value per lane)
template <class P, bool SWAP>
__global__ __launch_bounds__(BLOCK) void k_cond(const int* d, spint* g, spint* f, size_t n,
                                                Ld ldg, Ld ldf) {
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (size_t)gridDim.x * BLOCK) {
        spint x[1][P::N], y[1][P::N];
        load_soa<P, 1>(g, ldg, t, x);
        load_soa<P, 1>(f, ldf, t, y);
        const int b = d[t];
        if constexpr (SWAP) {
            Field<P>::modcsw(b, x[0], y[0]);
            store_soa<P, 1>(g, ldg, t, x);
        } else {
            Field<P>::modcmv(b, x[0], y[0]);
        }
        store_soa<P, 1>(f, ldf, t, y);
    }
}

// in-place normalisers / predicates; KIND selects the function, optional int result per element
enum { K_MODFSB = 0, K_FLATTEN, K_MODIS1, K_MODIS0, K_MODSIGN, K_MODHAF, K_MODQR, K_MODLIMBS, K_PROP };
template <class P, int KIND>
__global__ __launch_bounds__(BLOCK) void k_inplace(spint* a, int* out, size_t n, Ld ld) {
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (size_t)gridDim.x * BLOCK) {
        spint x[1][P::N];
        load_soa<P, 1>(a, ld, t, x);
        int r = 0;
        bool wr = false;
        if constexpr (KIND == K_MODFSB) { r = (int)Field<P>::modfsb(x[0]); wr = true; }
        if constexpr (KIND == K_FLATTEN) { r = (int)Field<P>::flatten(x[0]); wr = true; }
        if constexpr (KIND == K_PROP) { r = (int)Field<P>::prop(x[0]); wr = true; }        // the mask: -1 (all ones) or 0
        if constexpr (KIND == K_MODIS1) r = Field<P>::modis1(x[0]);
        if constexpr (KIND == K_MODIS0) r = Field<P>::modis0(x[0]);
        if constexpr (KIND == K_MODSIGN) r = Field<P>::modsign(x[0]);
        if constexpr (KIND == K_MODLIMBS) r = in_limb_budget<P>(x[0]) ? 1 : 0;      // every limb < 2^(Radix+2) (FOLD52 primes: <= (2^64-1)/mm)
        if constexpr (KIND == K_MODHAF) { Field<P>::modhaf(x[0]); wr = true; }
        if constexpr (KIND == K_MODQR) {
            bool fast = false;
            if constexpr (P::SPLIT > 0) fast = __all(in_split_contract<P>(x[0]));
            r = fast ? Field<P, true>::modqr(nullptr, x[0]) : Field<P, false>::modqr(nullptr, x[0]);
        }
        if (wr) store_soa<P, 1>(a, ld, t, x);
        if (out) out[t] = r;
    }
}

template <class P>
__global__ __launch_bounds__(BLOCK) void k_cmp(const spint* a, const spint* b, int* out,
                                               size_t n, Ld lda, Ld ldb) {
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (size_t)gridDim.x * BLOCK) {
        spint x[1][P::N], y[1][P::N];
        load_soa<P, 1>(a, lda, t, x);
        load_soa<P, 1>(b, ldb, t, y);
        out[t] = Field<P>::modcmp(x[0], y[0]);
    }
}

// shifts by less than a word (modshl / modshr), in place; shr returns the shifted-out bits
template <class P, bool LEFT>
__global__ __launch_bounds__(BLOCK) void k_shift(unsigned k, spint* a, int* out, size_t n, Ld ld) {
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (size_t)gridDim.x * BLOCK) {
        spint x[1][P::N];
        load_soa<P, 1>(a, ld, t, x);
        int r = 0;
        if constexpr (LEFT) Field<P>::modshl(k, x[0]); else r = Field<P>::modshr(k, x[0]);
        store_soa<P, 1>(a, ld, t, x);
        if (out) out[t] = r;
    }
}

// fill with a constant element: modzer / modone / modint(x) / mod2r(r)
enum { K_INT = 0, K_2R };
template <class P, int KIND>
__global__ __launch_bounds__(BLOCK) void k_fill(int val, spint* a, size_t n, Ld ld) {
    spint x[1][P::N];
    if constexpr (KIND == K_INT) {
        if (val == 0) Field<P>::modzer(x[0]); else Field<P>::modint(val, x[0]);
    } else {
        Field<P>::mod2r((unsigned)val, x[0]);
    }
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (size_t)gridDim.x * BLOCK)
        store_soa<P, 1>(a, ld, t, x);
}

// bytes <-> limbs.  AoS records of NBYTES big-endian bytes (what modimp / modexp take), one record per
// lane.  When NBYTES is a multiple of 8 the record moves as 64-bit words (word k of the integer = the
// byte-swapped chunk NW-1-k); otherwise (e.g. 66-byte NIST521 records) bytes are assembled one by one.
template <class P>
__device__ __forceinline__ void load_be_record(const unsigned char* bytes, size_t t, spint* w) {
    constexpr int NW = Field<P>::NW, NB = P::NBYTES;
    if constexpr (NB % 8 == 0) {
        const spint* src = reinterpret_cast<const spint*>(bytes) + t * NW;
        static_for<0, NW>([&](auto K) { w[K] = __builtin_bswap64(src[NW - 1 - K]); });
    } else {
        const unsigned char* src = bytes + t * NB;
        static_for<0, NW>([&](auto K) { w[K] = 0; });
        static_for<0, NB>([&](auto B) {
            constexpr int pos = NB - 1 - B;                 // byte significance (0 = least)
            w[pos / 8] |= (spint)src[B] << (8 * (pos % 8));
        });
    }
}
template <class P>
__device__ __forceinline__ void store_be_record(unsigned char* bytes, size_t t, const spint* w) {
    constexpr int NW = Field<P>::NW, NB = P::NBYTES;
    if constexpr (NB % 8 == 0) {
        spint* dst = reinterpret_cast<spint*>(bytes) + t * NW;
        static_for<0, NW>([&](auto K) { dst[NW - 1 - K] = __builtin_bswap64(w[K]); });
    } else {
        unsigned char* dst = bytes + t * NB;
        static_for<0, NB>([&](auto B) {
            constexpr int pos = NB - 1 - B;
            dst[B] = (unsigned char)(w[pos / 8] >> (8 * (pos % 8)));
        });
    }
}
template <class P>
__global__ __launch_bounds__(BLOCK) void k_imp(const unsigned char* bytes, spint* a, int* flag, size_t n, Ld ld) {
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (size_t)gridDim.x * BLOCK) {
        spint w[Field<P>::NW];
        load_be_record<P>(bytes, t, w);
        spint x[1][P::N];
        int r = Field<P>::modimp_words(w, x[0]);
        store_soa<P, 1>(a, ld, t, x);
        if (flag) flag[t] = r;
    }
}
template <class P>
__global__ __launch_bounds__(BLOCK) void k_exp(const spint* a, unsigned char* bytes, size_t n, Ld ld) {
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (size_t)gridDim.x * BLOCK) {
        spint x[1][P::N];
        load_soa<P, 1>(a, ld, t, x);
        spint w[Field<P>::NW];
        Field<P>::modexp_words(x[0], w);
        store_be_record<P>(bytes, t, w);
    }
}

// Synthetic field elements for benchmarks and full-size tests (SURVEY 8(d) input recipe): a splitmix64 stream keyed by
// (seed, array id); element j takes the NWD = ceil(Nbits/64)+1 consecutive outputs number j*NWD+1 .. j*NWD+NWD as a
// little-endian integer (bias below 2^-64) and reduces it mod p with field calls only -- Horner over the words with
// modmul by nres(2^64) (keeps plain values plain) and modadd, then modfsb -- leaving the CANONICAL limbs of a value
// uniform in [0,p) (plain, not nres'd).  plus_p: the same value + p, top limb unmasked: a representative in [p,2p).
// Regenerable on the host from (seed, array, j) alone (tests/util.py uniform_model).
MA_DEV spint splitmix64_at(spint s0, spint t) {
    spint z = s0 + (t + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
template <class P>
__global__ __launch_bounds__(BLOCK) void k_uniform(spint s0, size_t first, int plus_p, spint* out, size_t n, Ld ld) {
    using F = Field<P>;
    constexpr int NWD = (P::NBITS + 63) / 64 + 1;
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (size_t)gridDim.x * BLOCK) {
        const spint base = (spint)(first + t) * (spint)NWD;
        spint c[P::N], e[P::N], acc[1][P::N];
        F::mod2r(64, c);
        auto word_elem = [&](int k, spint* x) {                     // one 64-bit word as plain limbs
            spint w = splitmix64_at(s0, base + (spint)k);
            x[0] = w & F::MASK;
            x[1] = w >> P::RADIX;
            static_for<2, P::N>([&](auto I) { x[I] = 0; });
        };
        word_elem(NWD - 1, acc[0]);
#pragma unroll 1
        for (int k = NWD - 2; k >= 0; k--) {
            F::modmul(acc[0], c, acc[0]);
            word_elem(k, e);
            F::modadd(acc[0], e, acc[0]);
        }
        (void)F::modfsb(acc[0]);
        if (plus_p) {
            F::template addp<1>(acc[0], ~(spint)0);
            (void)F::prop(acc[0]);
        }
        store_soa<P, 1>(out, ld, t, acc);
    }
}

// The reference's timing protocol on the GPU (time.c: pseudo.py:1177-1386; its CUDA form
// simd/pseudo_cuda.py:1163-1231 runs the same dependent chains inside one thread): every lane runs the
// serially dependent chain on its own operands, entirely in registers, and leaves redc(z).
// KIND 0: `outer` x 200 x 5 modmul;  1: `outer` x 500 x 2 modsqr;  2: `outer` x 2 modinv.
template <class F, class P, int KIND>
MA_DEV void time_chain(spint* x, spint* y, spint* z, long outer) {
    // long fields: one out-of-line copy of the product / squaring (field.h chain_mul, chain_nsqr) instead of five / two inlined ones per
    // policy -- k_time<SIDH610, 0, 2> stood at 329 registers with both policies' chains inlined
    constexpr bool OOL = P::N >= 9;
    auto mul = [](const spint* a, const spint* b, spint* c) { if constexpr (OOL) F::chain_mul(a, b, c); else F::modmul(a, b, c); };
    F::nres(x, x);
    if constexpr (KIND == 0) {
        F::nres(y, y);
#pragma unroll 1
        for (long i = 0; i < outer * 200; i++) {
            mul(x, y, z);
            mul(z, x, y);
            mul(y, z, x);
            mul(x, y, z);
            mul(z, x, y);
        }
    } else if constexpr (KIND == 1) {
#pragma unroll 1
        for (long i = 0; i < outer * 500; i++) {
            if constexpr (OOL) {
                F::modcpy(x, z);
                F::chain_nsqr(z, 1);
                F::modcpy(z, x);
                F::chain_nsqr(x, 1);
            } else {
                F::modsqr(x, z);
                F::modsqr(z, x);
            }
        }
    } else {
#pragma unroll 1
        for (long i = 0; i < outer; i++) {
            F::modinv(x, nullptr, z);
            F::modinv(z, nullptr, x);
        }
    }
    F::redc(z, z);
}
// POLICY 0: exact products; 1: split products, unguarded (MA_FORCE_FAST); 2: the wave vote of OpMulAuto, taken once on the
// operands -- the chain then stays inside the limb contract by closure (every link is a field-function output)
template <class P, int KIND, int POLICY = 0>
__global__ __launch_bounds__(BLOCK) void k_time(const spint* xs, const spint* ys, spint* zs, long outer, size_t n, Ld ld) {
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (size_t)gridDim.x * BLOCK) {
        spint x[1][P::N], y[1][P::N], z[1][P::N];
        load_soa<P, 1>(xs, ld, t, x);
        if constexpr (KIND == 0) load_soa<P, 1>(ys, ld, t, y);
        else static_for<0, P::N>([&](auto I) { y[0][I] = 0; });
        bool fast = POLICY == 1;
        if constexpr (POLICY == 2 && P::SPLIT > 0) fast = __all(in_split_contract<P>(x[0]) && in_split_contract<P>(y[0]));
        if constexpr (POLICY != 0 && P::SPLIT > 0) {
            if (fast) { time_chain<Field<P, true>, P, KIND>(x[0], y[0], z[0], outer); store_soa<P, 1>(zs, ld, t, z); continue; }
        }
        time_chain<Field<P, false>, P, KIND>(x[0], y[0], z[0], outer);
        store_soa<P, 1>(zs, ld, t, z);
    }
}

}  // namespace ma
