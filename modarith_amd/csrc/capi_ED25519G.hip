// modarith_amd/csrc/capi_ED25519G.hip -- ecn_ed25519_mulgen_get_batch: generator multiplication fused with the affine export
// (csrc/ed26.h ed25519_mulgen_get_one), the call sequence ecnXXXgen + ecnXXXmul + ecnXXXget that opens EdDSA key generation
// and signing in the reference (ed448.c:167-184, 196-199).  Fixed-base table: generated/comb_ED25519.h.
#include "../../include/modarith_amd.h"
#include "capi_common.h"
#include "generated/curve_ED25519.h"
#include "generated/comb_ED25519.h"
#include "kernels.h"
#include "ed26.h"
#include "ed26l_k.h"

namespace ma {

// COMB_ED25519_WINDOWS windows of COMB_ED25519_W bits x 2^(W-1) multiples x coordinates x limbs, the same for every lane: constant address space, wave-uniform indices
__constant__ int32_t comb_ed25519[] = { COMB_ED25519_VALUES };
struct CombED25519 {
    static constexpr int W = COMB_ED25519_W, NW = COMB_ED25519_WINDOWS;
    static __device__ __forceinline__ int32_t get(int idx) { return comb_ed25519[idx]; }
};

// e*G through the fixed-base table, ONE scalar per lane.  SELF = false (the product path, round 5): the Edwards (X : Y : Z) go to the shared
// inversion of csrc/edlad_k.h (one inversion per up to 32 records; rounds 2-4 shared one between the FOUR scalars of a lane).  SELF = true:
// the inversion in the kernel, for callers without scratch (a stream under capture).
template <bool SELF>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 4)))
void k_ed25519_mulgen(const unsigned char* e, unsigned char* xb, unsigned char* yb, int* sign, size_t first, size_t n, Ed26lWs ws) {
    using P = P_X25519;
    using F = Fe26;
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= n) return;
    Ed26<C_ED25519>::Ext R;
    {
        spint ew[4];
        load_be_record<P>(e, first + t, ew);
        ed25519_mulgen_acc<C_ED25519, CombED25519>(ew, R);
    }
    if constexpr (SELF) {
        uint32_t zi[10], ax[10];
        spint w[4];
        F::invert(R.Z, zi);
        F::mul(R.X, zi, ax);
        F::to_words(ax, w);
        const int sx = (int)(w[0] & 1);
        if (xb) store_be_record<P>(xb, first + t, w);
        F::mul(R.Y, zi, ax);
        F::to_words(ax, w);
        if (yb) store_be_record<P>(yb, first + t, w);
        if (sign) sign[first + t] = !yb ? (int)(w[0] & 1) : (!xb ? sx : 0);
    } else {
        ws.store_xyz(t, R.X, R.Y, R.Z);
    }
}

// rfc7748() on the base point u = 9 (ed26.h x25519_base_many): little-endian 32-byte records as rfc7748_X25519_batch takes them; MULGEN_G
// keys per lane (elements t, t + lanes, ... of a MULGEN_G * lanes stride) share one inversion.  (The one-key-per-lane form with the
// shared inversion of edlad_k.h, as k_ed25519_mulgen and k_x448_base run, was built for this kernel too in round 5 and left out: the
// compiler brings its window loop out at 168 registers + 101 spilled where the identical loop of k_ed25519_mulgen takes 129.)
#ifndef MULGEN_G
#define MULGEN_G 4
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 4)))
void k_x25519_base(const uint64_t* bk, uint64_t* bv, size_t n) {
    const size_t lanes = (size_t)gridDim.x * blockDim.x;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += MULGEN_G * lanes) {
        uint64_t ow[MULGEN_G][4];
        x25519_base_many<C_ED25519, CombED25519, MULGEN_G>(
            [&](int g, uint64_t* kw) {
                const size_t tg = t + (size_t)g * lanes, ts = tg < n ? tg : t;
                static_for<0, 4>([&](auto K) { kw[K] = bk[ts * 4 + K]; });
            }, ow);
        static_for<0, MULGEN_G>([&](auto GI) {
            const size_t tg = t + (size_t)GI * lanes;
            if (tg < n) static_for<0, 4>([&](auto K) { bv[tg * 4 + K] = ow[GI][K]; });
        });
    }
}

// round 5, the ladder form (csrc/ed26l.h): f*Q by the Montgomery ladder with the recovered Edwards point in extended coordinates,
// e*G added through the constant table; the inversions in front and behind are shared (ed26l_k.h)
// 163 VGPRs, no scratch: three waves per SIMD (a 128-register build for four waves measured the same rate, 1.128e8 mul_get/s, and
// spilled 39 registers in the recovery; two waves -- enforced from outside through an LDS claim -- lost 12 %: profiles/history/r05_lad_ab.log)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_ed25519_lad_gen2(const unsigned char* e, const unsigned char* f, size_t first, Ed26lWs ws) {
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= ws.m) return;
    using L = Ed26Lad<C_ED25519>;
    uint32_t u[10], w[10], x2[10], z2[10], x3[10], z3[10];
    bool f_odd;
    {
        spint fw[4];
        load_be_record<P_X25519>(f, first + t, fw);
        f_odd = (fw[0] & 1) != 0;
        ws.load_u(t, u);
        L::ladder(fw, u, x2, z2, x3, z3);
    }
    ws.load_u(t, u);
    const uint32_t fl = ws.load_w(t, w);
    Ed26<C_ED25519>::Ext R;
    L::recover<true>(u, w, fl, f_odd, x2, z2, x3, z3, R);
    spint ew[4];
    load_be_record<P_X25519>(e, first + t, ew);
    ed25519_mulgen_acc<C_ED25519, CombED25519, false>(ew, R);
    ws.store_xyz(t, R.X, R.Y, R.Z);
}

}  // namespace ma

using namespace ma;

extern "C" size_t ecn_ed25519_mulgen2_get_workspace_bytes(size_t n) { return ed26l_workspace_bytes(n); }

extern "C" int ecn_ed25519_mulgen2_get_batch(const char* e, const char* f, const ma_spint* Q, char* x, char* y, int* sign, size_t n, size_t ld,
                                             void* workspace, size_t workspace_bytes, void* st) {
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(f) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 7u) {
        set_error("ecn mulgen2_get: byte records must be 8-byte aligned");
        return (int)hipErrorInvalidValue;
    }
    hipStream_t s = (hipStream_t)st;
    EdLadScratch ws(workspace, workspace_bytes, ed26l_workspace_bytes(n), 8, s);
    if (!ws.p) {
        set_error(std::string("ecn mulgen2_get: no usable workspace -- " + std::string(ws.why) + " (pass ecn_ed25519_mulgen2_get_workspace_bytes(n) bytes; the library's own scratch pool is not available while the stream is being captured)"));
        return (int)hipErrorInvalidValue;
    }
    const unsigned char *eb = reinterpret_cast<const unsigned char*>(e), *fb = reinterpret_cast<const unsigned char*>(f);
    edlad_pipeline<LadT25519, 2>(Q, ld, reinterpret_cast<unsigned char*>(x), reinterpret_cast<unsigned char*>(y), sign, n, ws.p, s,
                                 [&](size_t first, size_t m, const Ed26lWs& w) { k_ed25519_lad_gen2<<<(unsigned)((m + 63) / 64), 64, 0, s>>>(eb, fb, first, w); });
    return check_launch("ecn mulgen2_get (ladder form)");
}

extern "C" int ecn_ed25519_mulgen_get_batch(const char* e, char* x, char* y, int* sign, size_t n, void* st) {
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 7u) {
        set_error("ecn mulgen_get: byte records must be 8-byte aligned");
        return (int)hipErrorInvalidValue;
    }
    hipStream_t s = (hipStream_t)st;
    const unsigned char* eb = reinterpret_cast<const unsigned char*>(e);
    unsigned char *xb = reinterpret_cast<unsigned char*>(x), *yb = reinterpret_cast<unsigned char*>(y);
    EdLadScratch ws(nullptr, 0, ed26l_workspace_bytes(n), 8, s);          // this entry point has no workspace argument: the library's scratch pool
    for (size_t first = 0; first < n; first += EDLAD_CHUNK) {
        const size_t m = n - first < EDLAD_CHUNK ? n - first : EDLAD_CHUNK;
        if (ws.p) {
            Ed26lWs w(ws.p, m);
            k_ed25519_mulgen<false><<<(unsigned)((m + 63) / 64), 64, 0, s>>>(eb, xb, yb, sign, first, m, w);
            edlad_export<LadT25519, 4>(w, xb, yb, sign, first, s);
        } else {
            k_ed25519_mulgen<true><<<(unsigned)((m + 63) / 64), 64, 0, s>>>(eb, xb, yb, sign, first, m, Ed26lWs(nullptr, m));
        }
    }
    return check_launch("ecn mulgen_get");
}

// bv = [bk](9): rfc7748(bk, base, bv) of rfc7748.c:297-333 for a batch of private keys, on the fixed-base table
extern "C" int rfc7748_X25519_base_batch(const char* bk, char* bv, size_t n, void* st) {
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(bk) | reinterpret_cast<uintptr_t>(bv)) & 7u) {
        set_error("rfc7748 base: byte records must be 8-byte aligned");
        return (int)hipErrorInvalidValue;
    }
    const size_t lanes = ((n + MULGEN_G - 1) / MULGEN_G + 63) / 64 * 64, cap = (size_t)4 * 1024 * 64;
    k_x25519_base<<<(unsigned)((lanes < cap ? lanes : cap) / 64), 64, 0, (hipStream_t)st>>>(
        reinterpret_cast<const uint64_t*>(bk), reinterpret_cast<uint64_t*>(bv), n);
    return check_launch("rfc7748 base");
}
