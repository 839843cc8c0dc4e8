// modarith_amd/csrc/ladder.h -- batched RFC 7748 Montgomery ladder for gfx950, one scalar
// multiplication per lane with the whole state in VGPRs.
//
// Per element this is the call sequence of the reference's rfc7748() (rfc7748.c:156-256, TWIST_SECURE
// branch): clamp, import u, 255/448 ladder steps (2 cswap, 4 add, 4 sub, 4 sqr, 5 mul, 1 mli), final
// cswap, modpro + modinv, modmul, export.  The field is used in its generic=False form, as
// rfc7748.c:20 prescribes.  Records are the contiguous AoS layout of the reference's CUDA kernel
// (simd/rfc7748_simt.cu:165-168,224): bk[j*Nbytes + i], RFC little-endian bytes; they are moved as
// 64-bit words.  Conditional swaps are lane-predicated selects (v_cndmask): there is no branch or
// address that depends on scalar bits.  The kernel is bound by 32-bit integer multiply-add issue,
// not by HBM (96 bytes of traffic per ~3*10^5 multiply-adds).
#pragma once
#include "field.h"

namespace ma {

template <class P, int A24, int COF>
__global__ __launch_bounds__(256) void k_rfc7748(const spint* bk, const spint* bu, spint* bv, size_t n) {
    using F = Field<P, true>;   // FAST product path where available: every input of a field call is in contract here
    constexpr int N = P::N, NW = F::NW, NBITS = P::NBITS;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        spint kw[NW], uw[NW];
        static_for<0, NW>([&](auto K) { kw[K] = bk[t * NW + K]; });
        static_for<0, NW>([&](auto K) { uw[K] = bu[t * NW + K]; });
        // mask the unused top bits of u (rfc7748.c:171-172, mask() 148-152)
        if constexpr (NBITS % 64 != 0) uw[NW - 1] &= (((spint)1 << (NBITS % 64)) - 1);
        // clamp (rfc7748.c:135-141): clear the cofactor bits, clear bits >= NBITS, set bit NBITS-1
        kw[0] &= ~(((spint)1 << COF) - 1);
        if constexpr (NBITS % 64 != 0) kw[NW - 1] &= (((spint)1 << (NBITS % 64)) - 1);
        kw[NW - 1] |= (spint)1 << ((NBITS - 1) % 64);

        spint u[N], x1[N], x2[N], z2[N], x3[N], z3[N];
        (void)F::modimp_words(uw, u);
        F::modcpy(u, x1);
        F::modone(x2);
        F::modzer(z2);
        F::modcpy(u, x3);
        F::modone(z3);

        // left-align the scalar so that bit NBITS-1 sits in the sign position of the top word; each
        // step reads that bit and shifts the multi-word scalar left by one (static register indices
        // only, one copy of the step body in the instruction stream)
        constexpr int LSH = 64 * NW - NBITS;
        if constexpr (LSH > 0) {
            static_for<0, NW>([&](auto KK) {
                constexpr int k = NW - 1 - KK;
                kw[k] <<= LSH;
                if constexpr (k > 0) kw[k] |= kw[k - 1] >> (64 - LSH);
            });
        }
        int swap = 0;
#pragma unroll 1
        for (int step = 0; step < NBITS; step++) {
            const int kt = (int)(kw[NW - 1] >> 63);
            static_for<0, NW>([&](auto KK) {
                constexpr int k = NW - 1 - KK;
                kw[k] <<= 1;
                if constexpr (k > 0) kw[k] |= kw[k - 1] >> 63;
            });
            swap ^= kt;
            F::modcsw(swap, x2, x3);
            F::modcsw(swap, z2, z3);
            swap = kt;
            spint A[N], B[N], C[N], D[N], AA[N], BB[N], E[N];
            F::modadd_lazy(x2, z2, A);
            F::modadd_lazy(x3, z3, C);
            F::modsub_lazy(x2, z2, B);
            F::modsub_lazy(x3, z3, D);
            F::modsqr(A, AA);
            F::modsqr(B, BB);
            F::modmul(D, A, D);
            F::modmul(C, B, C);
            F::modsub_lazy(D, C, z3);
            F::modsub_lazy(AA, BB, E);
            F::modmli(E, A24, z2);
            F::modadd_lazy(D, C, x3);
            F::modadd_lazy(z2, AA, z2);
            F::modmul(z2, E, z2);
            F::modsqr(x3, x3);
            F::modsqr(z3, z3);
            F::modmul(z3, x1, z3);
            F::modmul(AA, BB, x2);
        }
        F::modcsw(swap, x2, x3);
        F::modcsw(swap, z2, z3);

        spint h[N];
        F::modpro(z2, h);
        F::modinv(z2, h, z2);
        F::modmul(x2, z2, x2);
        spint ow[NW];
        F::modexp_words(x2, ow);
        static_for<0, NW>([&](auto K) { bv[t * NW + K] = ow[K]; });
    }
}

}  // namespace ma
