// modarith_amd/csrc/wn_affine.h -- AFFINE window tables for the fused P-256 kernels (round 5): the multiples 1P .. 8P of every record
// are brought to Z = 1 with one inversion per up to 32 table entries, so that the window loop adds them with the Jacobian MIXED addition
// (7M + 4S = 1 360 multiply-adds against the 2 035 of the general one): 64 additions per scalar at 675 multiply-adds less, for about
// 1 000 per table entry (8 000 per record) in the kernel below -- mul_get 5.2 -> 5.9-6.1e7/s.  (secp256k1 stays on projective tables:
// its complete mixed addition, 11M = 1 310 multiply-adds, saves too little against the complete addition's 12M = 1 433 to pay for the
// normalisation; built and measured, 8.3 -> 8.4e7/s for mul_get, 6.2 -> 5.9e7/s for mulgen2_get.)
//
// The table is per RECORD of a chunk, not per resident lane: T[entry][word][record] (wn26.h's put / get with the record count as the
// stride), 15 words per entry while it is projective, the first 10 (x, y) after k_wn_table_affine; C holds the prefix products of the
// shared inversion, flag[record] = "P is the point at infinity" (then every multiple is, and the window kernels never use the
// entries).  An entry with Z = 0 counts as 1 in the shared inversion in BOTH passes, whatever the flag says: a record whose point is
// not on the curve cannot disturb the records it shares an inversion with.  1 284 bytes per record (2 564 with two tables) for at most WNAFF_CHUNK records.  Pipeline per chunk: the table kernel of the curve (wj26.h) ->
// k_wn_table_affine -> the window kernel -> wn_export.h.
//
// JAC = true: entries are Jacobian (x = X / Z^2, y = Y / Z^3); false: homogeneous (x = X / Z, y = Y / Z).
#pragma once
#include "capi_common.h"
#include "kernels.h"

namespace ma {

constexpr size_t WNAFF_CHUNK = (size_t)1 << 19;

struct WnAffWs {
    uint64_t *T, *C;
    uint32_t* flag;
    size_t m;                               // records of this chunk = row length
    int ne;                                 // entries per record: 8 (one table) or 16 (two: e P + f Q)
    static size_t bytes_per_record(int ne_) { return (size_t)(ne_ * 15 + ne_ * 5) * sizeof(uint64_t) + sizeof(uint32_t); }
    static size_t bytes(size_t n, int ne_ = 8) { return ((n < WNAFF_CHUNK ? n : WNAFF_CHUNK) * bytes_per_record(ne_) + 7) & ~(size_t)7; }    // (what follows it in a workspace stays 8-byte aligned)
    WnAffWs(void* ws, size_t m_, int ne_ = 8) : m(m_), ne(ne_) {
        T = reinterpret_cast<uint64_t*>(ws);
        C = T + (size_t)ne * 15 * m;
        flag = reinterpret_cast<uint32_t*>(C + (size_t)ne * 5 * m);
    }
    // limbs of coordinate c (0: X / x, 1: Y / y, 2: Z) of entry e of record t
    template <class F>
    MA_DEV void load(int e, int c, size_t t, int32_t* f) const {
        uint64_t w[5];
        static_for<0, 5>([&](auto K) { w[K] = T[(size_t)(e * 15 + c * 5 + K) * m + t]; });
        F::unpack(w, f);
    }
    template <class F>
    MA_DEV void store(int e, int c, size_t t, const int32_t* f) const {
        uint64_t w[5];
        F::pack(f, w);
        static_for<0, 5>([&](auto K) { T[(size_t)(e * 15 + c * 5 + K) * m + t] = w[K]; });
    }
};

// lane j of L: the ne entries of records j, j + L, ..., j + (R - 1) L -- up to 32 entries under one inversion.  flag[t]: bit 0 = the
// first table's point is at infinity (its entry 1 has Z = 0; then all its multiples have), bit 1 = the second table's (entries 8..15)
template <class F, bool JAC>
MA_DEV void wn_table_affine_lane(const WnAffWs& ws, size_t L, int R, size_t j) {
    int32_t c[10], z[10], one[10];
    F::set_one(one);
    F::set_one(c);
    const int ne = ws.ne;
#pragma unroll 1
    for (int g = 0; g < ne * R; g++) {
        const int e = g % ne;
        const size_t t = (size_t)(g / ne) * L + j;
        if (t >= ws.m) continue;
        ws.load<F>(e, 2, t, z);
        uint64_t zw[4];
        F::to_words(z, zw);
        const bool z0 = (zw[0] | zw[1] | zw[2] | zw[3]) == 0;
        if (e == 0) ws.flag[t] = z0 ? 1u : 0u;
        if (e == 8) ws.flag[t] |= z0 ? 2u : 0u;
        F::select(z0, z, one, z);
        F::mul(c, z, c);
        uint64_t w[5];
        F::pack(c, w);
        static_for<0, 5>([&](auto K) { ws.C[(size_t)(e * 5 + K) * ws.m + t] = w[K]; });
    }
    int32_t inv[10];
    F::invert(c, inv);
#pragma unroll 1
    for (int g = ne * R - 1; g >= 0; g--) {
        const int e = g % ne;
        const size_t t = (size_t)(g / ne) * L + j;
        if (t >= ws.m) continue;                                    // (counted as 1 above: nothing to undo)
        int32_t zi[10], x[10];
        ws.load<F>(e, 2, t, z);
        // the SAME test as on the way up, entry by entry: what counted as 1 there must count as 1 here.  flag[t] only says that entry 0 /
        // entry 8 is at infinity; an input off the curve -- (x, 0, Z != 0): its doubling has Z3 = 2 Y Z = 0 -- has Z = 0 in entries 2P, 4P,
        // 6P, 8P alone, and taking their Z = 0 into `inv` here zeroed the tables of every record EARLIER in this lane's column (round-5
        // advisor: one unvalidated public key in a verification batch spoiled up to three other records).  A record off the curve still
        // means nothing itself; it no longer touches its neighbours.
        uint64_t zw[4];
        F::to_words(z, zw);
        const bool z0 = (zw[0] | zw[1] | zw[2] | zw[3]) == 0;
        F::select(z0, z, one, z);
        if (g > 0) {
            const int ep = (g - 1) % ne;
            const size_t tp = (size_t)((g - 1) / ne) * L + j;
            uint64_t w[5];
            int32_t cp[10];
            static_for<0, 5>([&](auto K) { w[K] = ws.C[(size_t)(ep * 5 + K) * ws.m + tp]; });
            F::unpack(w, cp);
            F::mul(inv, cp, zi);
            F::mul(inv, z, inv);
        } else {
            F::copy(inv, zi);
        }
        if constexpr (JAC) {
            int32_t zi2[10];
            F::sqr(zi, zi2);
            ws.load<F>(e, 0, t, x);
            F::mul(x, zi2, x);
            ws.store<F>(e, 0, t, x);
            F::mul(zi2, zi, zi2);
            ws.load<F>(e, 1, t, x);
            F::mul(x, zi2, x);
            ws.store<F>(e, 1, t, x);
        } else {
            ws.load<F>(e, 0, t, x);
            F::mul(x, zi, x);
            ws.store<F>(e, 0, t, x);
            ws.load<F>(e, 1, t, x);
            F::mul(x, zi, x);
            ws.store<F>(e, 1, t, x);
        }
    }
}

template <class F, bool JAC, int TAG>
__global__ __launch_bounds__(64) void k_wn_table_affine(WnAffWs ws, size_t L, int R) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < L) wn_table_affine_lane<F, JAC>(ws, L, R, j);
}

template <class F, bool JAC, int TAG>
void wn_table_affine(const WnAffWs& ws, hipStream_t s) {
    size_t r = (ws.m + 65535) / 65536, rmax = (size_t)(32 / ws.ne);
    if (r > rmax) r = rmax;
    if (r < 1) r = 1;
    const size_t L = (ws.m + r - 1) / r;
    k_wn_table_affine<F, JAC, TAG><<<(unsigned)((L + 63) / 64), 64, 0, s>>>(ws, L, (int)r);
}

// +- entry base + |m| - 1 of record t's affine table (m = 1..8; 0 leaves (0, 0)), all eight entries from base on read (wn26.h lookup),
// the sign applied to y
template <class F>
MA_DEV void wn_affine_lookup(const WnAffWs& ws, size_t t, uint32_t m, bool neg, int32_t* sx, int32_t* sy, int base = 0) {
    uint64_t sel[10];
    static_for<0, 10>([&](auto K) { sel[K] = 0; });
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::: "memory");
#endif
#pragma unroll 1
    for (int e = 0; e < 8; e += 2) {
        uint64_t ent[2][10];
        static_for<0, 2>([&](auto EI) {
            static_for<0, 10>([&](auto K) { ent[EI][K] = ws.T[(size_t)((base + e + EI) * 15 + K) * ws.m + t]; });
        });
        static_for<0, 2>([&](auto EI) {
            const bool hit = (m == (uint32_t)(e + EI + 1));
            static_for<0, 10>([&](auto K) {
                const uint64_t a = ent[EI][K], b = sel[K];
                sel[K] = hit ? a : b;
            });
        });
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" ::: "memory");
#endif
    }
    int32_t ny[10];
    F::unpack(sel, sx);
    F::unpack(sel + 5, sy);
    F::neg(sy, ny);
    F::select(neg, sy, ny, sy);
}

}  // namespace ma
