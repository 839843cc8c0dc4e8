// modarith_amd/csrc/fe_finish.h -- the end of a BATCH of RFC 7748 ladders with one inversion per `rounds` scalars
// (Montgomery's simultaneous inversion), for the fused ladders of fe26.h (X25519) and fe28.h (X448).
//
// rfc7748() ends every scalar multiplication in modinv + modmul + modexp (rfc7748.c:225-254): for X25519 254 squarings and
// 11 multiplications, 7.7 % of the whole function (X448: 447 + 13, about the same share).  The batched entry point does not
// owe one inversion to each lane: a first kernel (k_x25519_fe26_xz / k_x448_fe28_xz) runs the ladders and leaves canonical
// x2 in the output record and canonical z2 in a workspace; the second (k_fe_finish) gives lane j the elements
// {r * L + j : r < rounds} (L lanes, so that every access of a wave is one coalesced row), multiplies their z up into
// prefix products, inverts the last one, and walks back with three multiplications per element.  A z2 of zero (u = 0, the
// low-order points of RFC 7748 section 7) would annihilate the whole product: it is replaced by 1 on the way in and its
// output -- x2 * 0^(p-2) = 0 in the reference -- is set to zero on the way out, by a per-lane mask (v_cndmask on the way in,
// v_and on the way out; field.h lane_mask()), never by a branch.  Same bytes as the one-inversion-per-lane kernels for every input.
#pragma once
#include "field.h"

namespace ma {

template <class F, int NL, int NW>
struct FeFinish {
    // z words of element e (canonical, as the first kernel left them), with 0 -> 1; returns the zero flag
    static MA_DEV bool load_z(const uint64_t* wz, size_t n, size_t e, bool valid, uint32_t* z) {
        uint64_t zw[NW];
        static_for<0, NW>([&](auto K) { zw[K] = (K == 0) ? 1 : 0; });
        if (valid) static_for<0, NW>([&](auto K) { zw[K] = wz[(size_t)K * n + e]; });
        uint64_t any = 0;
        static_for<0, NW>([&](auto K) { any |= zw[K]; });
        const bool zero = any == 0;
        zw[0] = zero ? 1 : zw[0];
        F::from_words(zw, z);
        return zero;
    }
    static MA_DEV void run(uint64_t* bv, const uint64_t* wz, uint32_t* wc, size_t n, size_t L, int rounds, size_t j) {
        uint32_t c[NL], z[NL];
        F::set(1, c);
#pragma unroll 1
        for (int r = 0; r < rounds; r++) {
            const size_t e = (size_t)r * L + j;
            const bool valid = e < n;
            (void)load_z(wz, n, e, valid, z);
            F::mul(c, z, c);
            if (valid) static_for<0, NL>([&](auto I) { wc[(size_t)I * n + e] = c[I]; });
        }
        uint32_t inv[NL];
        F::invert(c, inv);
#pragma unroll 1
        for (int r = rounds - 1; r >= 0; r--) {
            const size_t e = (size_t)r * L + j;
            if (e >= n) continue;                                   // (its z counted as 1: nothing to undo)
            uint32_t zinv[NL], x[NL];
            const uint64_t keep = lane_mask(!load_z(wz, n, e, true, z));     // (field.h: an opaque mask, not a select)
            if (r > 0) {
                uint32_t cp[NL];
                const size_t e1 = e - L;
                static_for<0, NL>([&](auto I) { cp[I] = wc[(size_t)I * n + e1]; });
                F::mul(inv, cp, zinv);
                F::mul(inv, z, inv);
            } else {
                F::copy(inv, zinv);
            }
            uint64_t xw[NW], ow[NW];
            static_for<0, NW>([&](auto K) { xw[K] = bv[e * NW + K]; });
            F::from_words(xw, x);
            F::mul(x, zinv, x);
            F::to_words(x, ow);
            static_for<0, NW>([&](auto K) { bv[e * NW + K] = ow[K] & keep; });
        }
    }
};

// one inversion per `rounds` elements; wc[NL][n] receives the prefix products (limb-major, 32-bit)
template <class F, int NL, int NW>
__global__ __launch_bounds__(256) void k_fe_finish(uint64_t* bv, const uint64_t* wz, uint32_t* wc, size_t n, size_t L, int rounds) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < L) FeFinish<F, NL, NW>::run(bv, wz, wc, n, L, rounds, j);
}

// ---- round 5: the same trick with TWO numerators per denominator, for the ladder form of the fused Edwards multiplications
// (ed26l.h: u = nu / D and w = nw / D in front of the ladder, x = X / Z and y = Y / Z behind it).  A, B, C: canonical words,
// word-major [NW][n] (every access of a wave is one coalesced row).  B[e] <- B[e] / A[e], C[e] <- C[e] / A[e] handed to a SINK:
//   sink(e, bw, cw)                 bw, cw = the canonical quotients (zero where A[e] was zero)
template <class F, int NL, int NW>
struct FeBatchDiv {
    using Z = FeFinish<F, NL, NW>;
    template <class SINK>
    static MA_DEV void run(const uint64_t* A, const uint64_t* B, const uint64_t* Cn, uint32_t* wc, size_t n, size_t L, int rounds, size_t j, SINK& sink) {
        uint32_t c[NL], z[NL];
        F::set(1, c);
#pragma unroll 1
        for (int r = 0; r < rounds; r++) {
            const size_t e = (size_t)r * L + j;
            const bool valid = e < n;
            (void)Z::load_z(A, n, e, valid, z);
            F::mul(c, z, c);
            if (valid) static_for<0, NL>([&](auto I) { wc[(size_t)I * n + e] = c[I]; });
        }
        uint32_t inv[NL];
        F::invert(c, inv);
#pragma unroll 1
        for (int r = rounds - 1; r >= 0; r--) {
            const size_t e = (size_t)r * L + j;
            if (e >= n) continue;                                   // (its denominator counted as 1: nothing to undo)
            uint32_t zinv[NL], x[NL];
            const uint64_t keep = lane_mask(!Z::load_z(A, n, e, true, z));
            if (r > 0) {
                uint32_t cp[NL];
                const size_t e1 = e - L;
                static_for<0, NL>([&](auto I) { cp[I] = wc[(size_t)I * n + e1]; });
                F::mul(inv, cp, zinv);
                F::mul(inv, z, inv);
            } else {
                F::copy(inv, zinv);
            }
            uint64_t xw[NW], bw[NW], cw[NW];
            static_for<0, NW>([&](auto K) { xw[K] = B[(size_t)K * n + e]; });
            F::from_words(xw, x);
            F::mul(x, zinv, x);
            F::to_words(x, bw);
            static_for<0, NW>([&](auto K) { xw[K] = Cn[(size_t)K * n + e]; });
            F::from_words(xw, x);
            F::mul(x, zinv, x);
            F::to_words(x, cw);
            static_for<0, NW>([&](auto K) { bw[K] &= keep; cw[K] &= keep; });
            sink(e, bw, cw);
        }
    }
};
// TAG keeps the instantiations of different translation units apart (one code object each)
template <class F, int NL, int NW, class SINK, int TAG>
__global__ __launch_bounds__(64) void k_fe_batch_div(const uint64_t* A, const uint64_t* B, const uint64_t* Cn, uint32_t* wc, size_t n, size_t L, int rounds, SINK sink) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < L) FeBatchDiv<F, NL, NW>::run(A, B, Cn, wc, n, L, rounds, j, sink);
}
// the quotients back into their own arrays as canonical words
template <int NW>
struct SinkWords {
    uint64_t *B, *Cn;
    size_t n;
    MA_DEV void operator()(size_t e, uint64_t* bw, uint64_t* cw) const {
        static_for<0, NW>([&](auto K) { B[(size_t)K * n + e] = bw[K]; Cn[(size_t)K * n + e] = cw[K]; });
    }
};

}  // namespace ma
