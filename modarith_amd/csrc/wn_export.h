// modarith_amd/csrc/wn_export.h -- the affine export of the fused Weierstrass kernels (P-256 on fm26.h, secp256k1 on fk26.h) with the
// inversion SHARED by up to 32 records (round 5), as the fused Edwards kernels share theirs (csrc/edlad_k.h).
//
// Rounds 2-4 ended every fused scalar multiplication with its own inversion of Z (255 squarings + 15 multiplications: 27 400
// multiply-adds on fm26, 21 700 on fk26 -- 6-8 % of the kernel).  Now the multiplication kernel leaves the homogeneous (X : Y : Z) of
// its result in the workspace -- the limbs as they are, two per 64-bit word, rows of the chunk so that a wave writes one coalesced row
// per word -- and k_wn_export runs Montgomery's trick down a column of up to 32 records per lane: prefix products of the Z (a Z that is
// zero mod p counts as 1 and its record leaves as x = 0, y = 1, the bytes ecnXXXget gives for the point at infinity,
// weierstrass.c:299-310), ONE inversion, and on the way back x = X / Z, y = Y / Z as canonical big-endian records.  About 1 800
// multiply-adds per record.  Same bytes as before for every input (tests/test_gpu_fused.py, test_gpu_weierstrass.py).
//
// Workspace: 160 bytes per record (X, Y, Z and the prefix product, five words each) for at most WNEXP_CHUNK records whatever the batch
// size; the entry points run the batch chunk by chunk.  It follows the window tables in the caller's workspace
// (ecn_<c>_*_get_workspace_bytes(n) covers both).
#pragma once
#include "capi_common.h"
#include "kernels.h"

namespace ma {

constexpr size_t WNEXP_CHUNK = (size_t)1 << 20;

struct WnExpWs {
    uint64_t *X, *Y, *Z, *C;
    size_t m;                               // records of this chunk = row length
    static constexpr size_t BYTES_PER_RECORD = 4 * 5 * sizeof(uint64_t);
    static size_t bytes(size_t n) { return (n < WNEXP_CHUNK ? n : WNEXP_CHUNK) * BYTES_PER_RECORD; }
    WnExpWs(void* ws, size_t m_) : m(m_) {
        X = reinterpret_cast<uint64_t*>(ws);
        Y = X + 5 * m;
        Z = Y + 5 * m;
        C = Z + 5 * m;
    }
    // record t of the chunk (limbs with |limb| < 2^31 as they come out of the last addition)
    template <class F>
    MA_DEV void store(size_t t, const int32_t* x, const int32_t* y, const int32_t* z) const {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(t));         // the row addresses are formed HERE, not above the caller's loops (edlad_k.h store_xyz)
#endif
        uint64_t w[5];
        F::pack(x, w);
        static_for<0, 5>([&](auto K) { X[(size_t)K * m + t] = w[K]; });
        F::pack(y, w);
        static_for<0, 5>([&](auto K) { Y[(size_t)K * m + t] = w[K]; });
        F::pack(z, w);
        static_for<0, 5>([&](auto K) { Z[(size_t)K * m + t] = w[K]; });
    }
    template <class F>
    MA_DEV void load(const uint64_t* row, size_t e, int32_t* f) const {
        uint64_t w[5];
        static_for<0, 5>([&](auto K) { w[K] = row[(size_t)K * m + e]; });
        F::unpack(w, f);
    }
};

// lane j of L: records j, j + L, j + 2L, ... of the chunk
template <class F, class P, int TAG>
__global__ __launch_bounds__(64) void k_wn_export(WnExpWs ws, size_t L, int rounds, unsigned char* xb, unsigned char* yb, int* sign, size_t first) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= L) return;
    int32_t c[10], z[10], one[10];
    F::set_one(one);
    F::set_one(c);
#pragma unroll 1
    for (int r = 0; r < rounds; r++) {
        const size_t e = (size_t)r * L + j;
        if (e >= ws.m) continue;
        ws.load<F>(ws.Z, e, z);
        uint64_t zw[4];
        F::to_words(z, zw);
        const bool z0 = (zw[0] | zw[1] | zw[2] | zw[3]) == 0;
        F::select(z0, z, one, z);
        F::mul(c, z, c);
        uint64_t w[5];
        F::pack(c, w);
        static_for<0, 5>([&](auto K) { ws.C[(size_t)K * ws.m + e] = w[K]; });
    }
    int32_t inv[10];
    F::invert(c, inv);
#pragma unroll 1
    for (int r = rounds - 1; r >= 0; r--) {
        const size_t e = (size_t)r * L + j;
        if (e >= ws.m) continue;                                    // (its denominator counted as 1: nothing to undo)
        int32_t zinv[10], x[10];
        ws.load<F>(ws.Z, e, z);
        uint64_t zw[4], xw[4], yw[4];
        F::to_words(z, zw);
        const bool z0 = (zw[0] | zw[1] | zw[2] | zw[3]) == 0;
        F::select(z0, z, one, z);
        if (r > 0) {
            int32_t cp[10];
            ws.load<F>(ws.C, e - L, cp);
            F::mul(inv, cp, zinv);
            F::mul(inv, z, inv);
        } else {
            F::copy(inv, zinv);
        }
        ws.load<F>(ws.X, e, x);
        F::mul(x, zinv, x);
        F::to_words(x, xw);
        ws.load<F>(ws.Y, e, x);
        F::mul(x, zinv, x);
        F::to_words(x, yw);
        const uint64_t keep = lane_mask(!z0);
        static_for<0, 4>([&](auto K) {
            xw[K] &= keep;
            yw[K] = (yw[K] & keep) | (K == 0 ? (1u & ~keep) : 0u);
        });
        const size_t t = first + e;
        if (xb) store_be_record<P>(xb, t, xw);
        if (yb) store_be_record<P>(yb, t, yw);
        if (sign) sign[t] = !yb ? (int)(yw[0] & 1) : (!xb ? (int)(xw[0] & 1) : 0);
    }
}

inline void wnexp_rounds(size_t m, size_t* L, int* rounds) {
    size_t r = (m + 65535) / 65536;
    if (r > 32) r = 32;
    if (r < 1) r = 1;
    *rounds = (int)r;
    *L = (m + r - 1) / r;
}
template <class F, class P, int TAG>
void wn_export(const WnExpWs& ws, unsigned char* x, unsigned char* y, int* sign, size_t first, hipStream_t s) {
    size_t L;
    int rounds;
    wnexp_rounds(ws.m, &L, &rounds);
    k_wn_export<F, P, TAG><<<(unsigned)((L + 63) / 64), 64, 0, s>>>(ws, L, rounds, x, y, sign, first);
}

}  // namespace ma
