// modarith_amd/csrc/ed26l_k.h -- the kernels of the ladder form of the fused ED25519 multiplications (csrc/ed26l.h) and the host
// side that queues them: prep -> shared inversion -> ladder (+ fixed-base part) -> shared inversion + export, CHUNK records at a time.
// Included by the units that own an entry point (capi_ED25519F.hip: mul_get; capi_ED25519G.hip: mulgen2_get); TAG keeps their
// kernel instantiations apart.
//
// Workspace per record (word-major rows of the chunk, every access of a wave one coalesced row): three canonical field elements
// A, B, C (4 x 64 bits each: D -> Z, nu -> u -> X, nw -> w -> Y), the prefix products of the shared inversions (10 x 32 bits) and one
// flag word: 140 bytes, for at most ED26L_CHUNK records whatever the batch size.
#pragma once
#include "capi_common.h"
#include "kernels.h"
#include "ed26l.h"

namespace ma {

constexpr size_t ED26L_CHUNK = (size_t)1 << 20;
constexpr size_t ED26L_BYTES_PER_RECORD = 3 * 4 * sizeof(uint64_t) + 10 * sizeof(uint32_t) + sizeof(uint32_t);
inline size_t ed26l_workspace_bytes(size_t n) { return (n < ED26L_CHUNK ? n : ED26L_CHUNK) * ED26L_BYTES_PER_RECORD; }

struct Ed26lWs {
    uint64_t *A, *B, *Cn;
    uint32_t *wc, *flags;
    size_t m;                       // records of this chunk = row length
    Ed26lWs(void* ws, size_t m_) : m(m_) {
        A = reinterpret_cast<uint64_t*>(ws);
        B = A + 4 * m;
        Cn = B + 4 * m;
        wc = reinterpret_cast<uint32_t*>(Cn + 4 * m);
        flags = wc + 10 * m;
    }
};

// the caller's workspace when it is large enough, else stream-ordered scratch of the library's own pool (released in stream order when
// this object goes); p = nullptr when neither is to be had (a stream under capture and no caller workspace)
struct Ed26lScratch {
    void* p = nullptr;
    void* own = nullptr;
    hipStream_t s;
    Ed26lScratch(void* workspace, size_t workspace_bytes, size_t n, hipStream_t s_) : s(s_) {
        const size_t need = ed26l_workspace_bytes(n);
        if (workspace && workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 7u) == 0) { p = workspace; return; }
        p = own = scratch_alloc(need, s);
    }
    ~Ed26lScratch() { if (own) scratch_free(own, s); }
};

// P = (X : Y : Z), rows of the caller's batch (limb stride ld), records first .. first + m
template <class C, int TAG>
__global__ __launch_bounds__(256) void k_ed26l_prep(const spint* Pb, size_t first, size_t ld, Ed26lWs ws) {
    using L = Ed26Lad<C>;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ws.m) return;
    spint X[5], Y[5], Z[5];
    static_for<0, 5>([&](auto I) {
        X[I] = Pb[(size_t)I * ld + first + t];
        Y[I] = Pb[(size_t)(5 + I) * ld + first + t];
        Z[I] = Pb[(size_t)(10 + I) * ld + first + t];
    });
    uint32_t D[10], nu[10], nw[10];
    const uint32_t fl = L::prep(X, Y, Z, D, nu, nw);
    uint64_t w[4];
    Fe26::to_words(D, w);
    static_for<0, 4>([&](auto K) { ws.A[(size_t)K * ws.m + t] = w[K]; });
    Fe26::to_words(nu, w);
    static_for<0, 4>([&](auto K) { ws.B[(size_t)K * ws.m + t] = w[K]; });
    Fe26::to_words(nw, w);
    static_for<0, 4>([&](auto K) { ws.Cn[(size_t)K * ws.m + t] = w[K]; });
    ws.flags[t] = fl;
}

// x = X / Z, y = Y / Z as the reference's big-endian records (ecnXXXget, edwards.c:221-239), records first .. of the caller's arrays
struct SinkExport25519 {
    unsigned char *xb, *yb;
    int* sign;
    size_t first;
    MA_DEV void operator()(size_t e, uint64_t* xw, uint64_t* yw) const {
        const size_t t = first + e;
        if (xb) store_be_record<P_X25519>(xb, t, xw);
        if (yb) store_be_record<P_X25519>(yb, t, yw);
        if (sign) sign[t] = !yb ? (int)(yw[0] & 1) : (!xb ? (int)(xw[0] & 1) : 0);
    }
};

inline void ed26l_rounds(size_t m, size_t* L, int* rounds) {
    size_t r = (m + 65535) / 65536;
    if (r > 32) r = 32;
    if (r < 1) r = 1;
    *rounds = (int)r;
    *L = (m + r - 1) / r;
}

// the two shared inversions around a ladder kernel `lad(first, m, ws)` (a callable that launches it), chunk by chunk
template <class C, int TAG, class LAD>
int ed26l_pipeline(const spint* P, size_t ld, unsigned char* x, unsigned char* y, int* sign, size_t n, void* workspace, hipStream_t s, LAD lad) {
    for (size_t first = 0; first < n; first += ED26L_CHUNK) {
        const size_t m = n - first < ED26L_CHUNK ? n - first : ED26L_CHUNK;
        Ed26lWs ws(workspace, m);
        size_t L;
        int rounds;
        ed26l_rounds(m, &L, &rounds);
        k_ed26l_prep<C, TAG><<<(unsigned)((m + 255) / 256), 256, 0, s>>>(P, first, ld, ws);
        k_fe_batch_div<Fe26, 10, 4, SinkWords<4>, TAG><<<(unsigned)((L + 63) / 64), 64, 0, s>>>(ws.A, ws.B, ws.Cn, ws.wc, m, L, rounds, SinkWords<4>{ws.B, ws.Cn, m});
        lad(first, m, ws);
        k_fe_batch_div<Fe26, 10, 4, SinkExport25519, TAG><<<(unsigned)((L + 63) / 64), 64, 0, s>>>(ws.A, ws.B, ws.Cn, ws.wc, m, L, rounds, SinkExport25519{x, y, sign, first});
    }
    return 0;
}

// the record of lane t of a chunk: u in front of the ladder, w and the flags behind it (nothing but the ladder's own state is live
// in its loop)
MA_DEV void ed26l_load_u(const Ed26lWs& ws, size_t t, uint32_t* u) {
    uint64_t uw[4];
    static_for<0, 4>([&](auto K) { uw[K] = ws.B[(size_t)K * ws.m + t]; });
    Fe26::from_words(uw, u);
}
MA_DEV uint32_t ed26l_load_w(const Ed26lWs& ws, size_t t, uint32_t* w) {
    uint64_t ww[4];
    static_for<0, 4>([&](auto K) { ww[K] = ws.Cn[(size_t)K * ws.m + t]; });
    Fe26::from_words(ww, w);
    return ws.flags[t];
}
MA_DEV void ed26l_store_xyz(const Ed26lWs& ws, size_t t, const uint32_t* X, const uint32_t* Y, const uint32_t* Z) {
    uint64_t w[4];
    Fe26::to_words(Z, w);
    static_for<0, 4>([&](auto K) { ws.A[(size_t)K * ws.m + t] = w[K]; });
    Fe26::to_words(X, w);
    static_for<0, 4>([&](auto K) { ws.B[(size_t)K * ws.m + t] = w[K]; });
    Fe26::to_words(Y, w);
    static_for<0, 4>([&](auto K) { ws.Cn[(size_t)K * ws.m + t] = w[K]; });
}

}  // namespace ma
