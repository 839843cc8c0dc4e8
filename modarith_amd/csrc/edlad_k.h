// modarith_amd/csrc/edlad_k.h -- the kernels of the ladder form of the fused Edwards multiplications (csrc/ed26l.h: ED25519 on fe26;
// csrc/ed28l.h: ED448 on fe28) and the host side that queues them: prep -> shared inversion -> ladder (+ fixed-base part) -> shared
// inversion + export, CHUNK records at a time.  Generic over a traits type T:
//     T::F (the 32-bit-limb field), T::P (the field.c prime: record size), T::NL / T::NW (limbs / 64-bit words of an element),
//     T::NIN (64-bit limbs of an input coordinate), T::prep(X, Y, Z, D, nu, nw) -> flags.
// Included by the units that own an entry point; TAG keeps their kernel instantiations apart.
//
// Workspace per record (word-major rows of the chunk, every access of a wave one coalesced row): three canonical field elements
// A, B, C (NW x 64 bits each: D -> Z, nu -> u -> X, nw -> w -> Y), the prefix products of the shared inversions (NL x 32 bits) and one
// flag word -- 140 bytes for ED25519, 236 for ED448 -- for at most EDLAD_CHUNK records whatever the batch size.
#pragma once
#include "capi_common.h"
#include "kernels.h"
#include "fe_finish.h"

namespace ma {

constexpr size_t EDLAD_CHUNK = (size_t)1 << 20;

template <class T>
struct EdLadWs {
    uint64_t *A, *B, *Cn;
    uint32_t *wc, *flags;
    size_t m;                       // records of this chunk = row length
    static constexpr size_t BYTES_PER_RECORD = 3 * T::NW * sizeof(uint64_t) + T::NL * sizeof(uint32_t) + sizeof(uint32_t);
    static size_t bytes(size_t n) { return (n < EDLAD_CHUNK ? n : EDLAD_CHUNK) * BYTES_PER_RECORD; }
    EdLadWs(void* ws, size_t m_) : m(m_) {
        A = reinterpret_cast<uint64_t*>(ws);
        B = A + T::NW * m;
        Cn = B + T::NW * m;
        wc = reinterpret_cast<uint32_t*>(Cn + T::NW * m);
        flags = wc + T::NL * m;
    }
    // the record of lane t of the chunk: u in front of the ladder, w and the flags behind it (nothing but the ladder's own state is
    // live in its loop); the Edwards (X : Y : Z) of the result for the second shared inversion
    MA_DEV void load_u(size_t t, uint32_t* u) const {
        uint64_t uw[T::NW];
        static_for<0, T::NW>([&](auto K) { uw[K] = B[(size_t)K * m + t]; });
        T::F::from_words(uw, u);
    }
    MA_DEV uint32_t load_w(size_t t, uint32_t* w) const {
        uint64_t ww[T::NW];
        static_for<0, T::NW>([&](auto K) { ww[K] = Cn[(size_t)K * m + t]; });
        T::F::from_words(ww, w);
        return flags[t];
    }
    // one numerator / denominator pair (B / A): the base-point ladders; the shared inversion is then given B for both of its numerators
    MA_DEV void store_nd(size_t t, const uint32_t* num, const uint32_t* den) const {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(t));
#endif
        uint64_t w[T::NW];
        T::F::to_words(den, w);
        static_for<0, T::NW>([&](auto K) { A[(size_t)K * m + t] = w[K]; });
        T::F::to_words(num, w);
        static_for<0, T::NW>([&](auto K) { B[(size_t)K * m + t] = w[K]; });
    }
    MA_DEV void store_xyz(size_t t, const uint32_t* X, const uint32_t* Y, const uint32_t* Z) const {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(t));     // the row addresses are formed HERE: hoisted above the caller's loops they cost 3 NW register pairs (spilled in k_ed448_lad_gen2)
#endif
        uint64_t w[T::NW];
        T::F::to_words(Z, w);
        static_for<0, T::NW>([&](auto K) { A[(size_t)K * m + t] = w[K]; });
        T::F::to_words(X, w);
        static_for<0, T::NW>([&](auto K) { B[(size_t)K * m + t] = w[K]; });
        T::F::to_words(Y, w);
        static_for<0, T::NW>([&](auto K) { Cn[(size_t)K * m + t] = w[K]; });
    }
};

// P = (X : Y : Z), rows of the caller's batch (limb stride ld), records first .. first + m
template <class T, int TAG>
__global__ __launch_bounds__(256) void k_edlad_prep(const spint* Pb, size_t first, size_t ld, EdLadWs<T> ws) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ws.m) return;
    spint X[T::NIN], Y[T::NIN], Z[T::NIN];
    static_for<0, T::NIN>([&](auto I) {
        X[I] = Pb[(size_t)I * ld + first + t];
        Y[I] = Pb[(size_t)(T::NIN + I) * ld + first + t];
        Z[I] = Pb[(size_t)(2 * T::NIN + I) * ld + first + t];
    });
    uint32_t D[T::NL], nu[T::NL], nw[T::NL];
    const uint32_t fl = T::prep(X, Y, Z, D, nu, nw);
    uint64_t w[T::NW];
    T::F::to_words(D, w);
    static_for<0, T::NW>([&](auto K) { ws.A[(size_t)K * ws.m + t] = w[K]; });
    T::F::to_words(nu, w);
    static_for<0, T::NW>([&](auto K) { ws.B[(size_t)K * ws.m + t] = w[K]; });
    T::F::to_words(nw, w);
    static_for<0, T::NW>([&](auto K) { ws.Cn[(size_t)K * ws.m + t] = w[K]; });
    ws.flags[t] = fl;
}

// x = X / Z, y = Y / Z as the reference's big-endian records (ecnXXXget, edwards.c:221-239), records first .. of the caller's arrays
template <class P>
struct SinkExportBE {
    unsigned char *xb, *yb;
    int* sign;
    size_t first;
    MA_DEV void operator()(size_t e, uint64_t* xw, uint64_t* yw) const {
        const size_t t = first + e;
        if (xb) store_be_record<P>(xb, t, xw);
        if (yb) store_be_record<P>(yb, t, yw);
        if (sign) sign[t] = !yb ? (int)(yw[0] & 1) : (!xb ? (int)(xw[0] & 1) : 0);
    }
};

inline void edlad_rounds(size_t m, size_t* L, int* rounds) {
    size_t r = (m + 65535) / 65536;
    if (r > 32) r = 32;
    if (r < 1) r = 1;
    *rounds = (int)r;
    *L = (m + r - 1) / r;
}

// the second shared inversion + export of a chunk
template <class T, int TAG>
void edlad_export(const EdLadWs<T>& ws, unsigned char* x, unsigned char* y, int* sign, size_t first, hipStream_t s) {
    size_t L;
    int rounds;
    edlad_rounds(ws.m, &L, &rounds);
    k_fe_batch_div<typename T::F, T::NL, T::NW, SinkExportBE<typename T::P>, TAG><<<(unsigned)((L + 63) / 64), 64, 0, s>>>(
        ws.A, ws.B, ws.Cn, ws.wc, ws.m, L, rounds, SinkExportBE<typename T::P>{x, y, sign, first});
}
// the two shared inversions around a ladder kernel `lad(first, m, ws)` (a callable that launches it), chunk by chunk
template <class T, int TAG, class LAD>
int edlad_pipeline(const spint* P, size_t ld, unsigned char* x, unsigned char* y, int* sign, size_t n, void* workspace, hipStream_t s, LAD lad) {
    for (size_t first = 0; first < n; first += EDLAD_CHUNK) {
        const size_t m = n - first < EDLAD_CHUNK ? n - first : EDLAD_CHUNK;
        EdLadWs<T> ws(workspace, m);
        size_t L;
        int rounds;
        edlad_rounds(m, &L, &rounds);
        k_edlad_prep<T, TAG><<<(unsigned)((m + 255) / 256), 256, 0, s>>>(P, first, ld, ws);
        k_fe_batch_div<typename T::F, T::NL, T::NW, SinkWords<T::NW>, TAG><<<(unsigned)((L + 63) / 64), 64, 0, s>>>(
            ws.A, ws.B, ws.Cn, ws.wc, m, L, rounds, SinkWords<T::NW>{ws.B, ws.Cn, m});
        lad(first, m, ws);
        edlad_export<T, TAG>(ws, x, y, sign, first, s);
    }
    return 0;
}

}  // namespace ma
