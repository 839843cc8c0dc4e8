// modarith_amd/csrc/capi_NIST256G.hip -- ecn_nist256_mulgen_get_batch: generator multiplication fused with the affine export
// (csrc/wn26.h wn26_mulgen_get_one), the call sequence ecnXXXgen + ecnXXXmul + ecnXXXget that opens key generation and
// signing in the reference's ECDSA code (nist256.c:150-161, 214-222).  Fixed-base table: generated/comb_NIST256.h.
#include "../../include/modarith_amd.h"
#include "capi_common.h"
#include "generated/curve_NIST256.h"
#include "generated/comb_NIST256.h"
#include "kernels.h"
#include "wn26.h"
#include "wj26.h"
#include "wn_export.h"

namespace ma {

// COMB_NIST256_WINDOWS windows of COMB_NIST256_W bits x 2^(W-1) multiples x coordinates x limbs, the same for every lane: constant address space, wave-uniform indices
__constant__ int32_t comb_nist256[] = { COMB_NIST256_VALUES };
struct CombNIST256 {
    static constexpr int W = COMB_NIST256_W, NW = COMB_NIST256_WINDOWS;
    static __device__ __forceinline__ int32_t get(int idx) { return comb_nist256[idx]; }
};

// e*G, ONE scalar per lane, (X : Y : Z) to the inversion shared by up to 32 records (csrc/wn_export.h; round 5).  Rounds 2-4 shared one
// inversion between the four scalars of a lane, whose four results cost 120 registers (the kernel below: 252-254 registers, two waves
// per SIMD); it stays for callers without scratch (a stream under capture: this entry point has no workspace argument).
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 4)))
void k_nist256_mulgen(const unsigned char* e, WnExpWs ex) {
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= ex.m) return;
    spint ew[4], kw[4];
    load_be_record<P_NIST256>(e, t, ew);
    Wj26::reduce_scalar(ew, kw);
    Wj26::Pt R;
    Wj26::mulgen_acc<CombNIST256>(kw, R);                   // Jacobian mixed additions (csrc/wj26.h)
    ex.store<Fm26>(t, R.X, R.Y, R.Z);
}

// MULGEN_G scalars per lane (elements t, t + lanes, ... of a MULGEN_G * lanes stride) share one inversion
#ifndef MULGEN_G
#define MULGEN_G 4
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 4)))
void k_nist256_mulgen_get(const unsigned char* e, unsigned char* xb, unsigned char* yb, int* sign, size_t n) {
    using P = P_NIST256;
    const size_t lanes = (size_t)gridDim.x * blockDim.x;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += MULGEN_G * lanes) {
        spint xw[MULGEN_G][4], yw[MULGEN_G][4];
        wn26_mulgen_get_many<CvNist256, CombNIST256, MULGEN_G>(
            [&](int g, spint* ew) { const size_t tg = t + (size_t)g * lanes; load_be_record<P>(e, tg < n ? tg : t, ew); }, xw, yw);
        static_for<0, MULGEN_G>([&](auto GI) {
            const size_t tg = t + (size_t)GI * lanes;
            if (tg < n) {
                if (xb) store_be_record<P>(xb, tg, xw[GI]);
                if (yb) store_be_record<P>(yb, tg, yw[GI]);
                if (sign) sign[tg] = !yb ? (int)(yw[GI][0] & 1) : (!xb ? (int)(xw[GI][0] & 1) : 0);
            }
        });
    }
}

// e*G + f*Q and its affine export (verification, nist256.c:251-256): f Q as in the mul_get unit (the table of Q brought to Z = 1,
// csrc/wn_affine.h; Jacobian mixed additions, csrc/wj26.h), the generator part through the constant table above
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_nist256_table2(const spint* Qb, size_t ld, WnAffWs ws) {
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= ws.m) return;
    Wj26::table_of([&](spint* X, spint* Y, spint* Z) {
        static_for<0, 5>([&](auto I) {
            X[I] = Qb[(size_t)I * ld + t];
            Y[I] = Qb[(size_t)(5 + I) * ld + t];
            Z[I] = Qb[(size_t)(10 + I) * ld + t];
        });
    }, ws, t);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_nist256_mulgen2_get(const unsigned char* e, const unsigned char* f, WnAffWs ws, WnExpWs ex) {
    using P = P_NIST256;
    using DIG = WnLds<4, 260>;
    __shared__ unsigned char digs[DIG::ROWS * 64];          // f's windows in LDS
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= ws.m) return;
    unsigned char* col = digs + threadIdx.x;
    {
        spint fw[4], kw[4];
        load_be_record<P>(f, t, fw);
        Wj26::reduce_scalar(fw, kw);
        DIG::fill(kw, col);
    }
    DIG dig{col};
    Wj26::Pt R;
    Wj26::mul_acc_aff(dig, ws, t, R);
    spint ew[4];
    load_be_record<P>(e, t, ew);                                // (fetched here: eight registers less across f Q)
    wn26_mulgen_acc<CvNist256, CombNIST256, false>(ew, R);
    ex.store<Fm26>(t, R.X, R.Y, R.Z);
}

}  // namespace ma

using namespace ma;

extern "C" size_t ecn_nist256_mulgen2_get_workspace_bytes(size_t n) { return WnAffWs::bytes(n) + WnExpWs::bytes(n); }

extern "C" int ecn_nist256_mulgen2_get_batch(const char* e, const char* f, const ma_spint* Q, char* x, char* y, int* sign, size_t n, size_t ld,
                                           void* workspace, size_t workspace_bytes, void* st) {
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(f) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 7u) {
        set_error("ecn mulgen2_get: byte records must be 8-byte aligned");
        return (int)hipErrorInvalidValue;
    }
    if (workspace == nullptr || (reinterpret_cast<uintptr_t>(workspace) & 7u) || workspace_bytes < ecn_nist256_mulgen2_get_workspace_bytes(n)) {
        set_error("ecn mulgen2_get: workspace missing, not 8-byte aligned or too small (see ecn_nist256_mulgen2_get_workspace_bytes)");
        return (int)hipErrorInvalidValue;
    }
    hipStream_t s = (hipStream_t)st;
    char* wsb = reinterpret_cast<char*>(workspace);
    for (size_t first = 0; first < n; first += WNAFF_CHUNK) {
        const size_t m = n - first < WNAFF_CHUNK ? n - first : WNAFF_CHUNK;
        const WnAffWs aw(wsb, m);
        const WnExpWs ex(wsb + WnAffWs::bytes(n), m);
        const unsigned g = (unsigned)((m + 63) / 64);
        k_nist256_table2<<<g, 64, 0, s>>>(Q + first, ld, aw);
        wn_table_affine<Fm26, true, 3>(aw, s);
        k_nist256_mulgen2_get<<<g, 64, 0, s>>>(reinterpret_cast<const unsigned char*>(e) + first * P_NIST256::NBYTES,
                                               reinterpret_cast<const unsigned char*>(f) + first * P_NIST256::NBYTES, aw, ex);
        wn_export<Fm26, P_NIST256, 3>(ex, reinterpret_cast<unsigned char*>(x), reinterpret_cast<unsigned char*>(y), sign, first, s);
    }
    return check_launch("ecn mulgen2_get");
}

extern "C" int ecn_nist256_mulgen_get_batch(const char* e, char* x, char* y, int* sign, size_t n, void* st) {
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 7u) {
        set_error("ecn mulgen_get: byte records must be 8-byte aligned");
        return (int)hipErrorInvalidValue;
    }
    hipStream_t s = (hipStream_t)st;
    EdLadScratch ws(nullptr, 0, WnExpWs::bytes(n), 8, s);              // no workspace argument: the library's scratch pool
    if (ws.p) {
        for (size_t first = 0; first < n; first += WNEXP_CHUNK) {
            const size_t m = n - first < WNEXP_CHUNK ? n - first : WNEXP_CHUNK;
            const WnExpWs ex(ws.p, m);
            k_nist256_mulgen<<<(unsigned)((m + 63) / 64), 64, 0, s>>>(reinterpret_cast<const unsigned char*>(e) + first * P_NIST256::NBYTES, ex);
            wn_export<Fm26, P_NIST256, 4>(ex, reinterpret_cast<unsigned char*>(x), reinterpret_cast<unsigned char*>(y), sign, first, s);
        }
        return check_launch("ecn mulgen_get");
    }
    const size_t lanes = ((n + MULGEN_G - 1) / MULGEN_G + 63) / 64 * 64, cap = (size_t)4 * 1024 * 64;       // MULGEN_G scalars per lane; at most 4 waves on each of the 1024 SIMDs
    k_nist256_mulgen_get<<<(unsigned)((lanes < cap ? lanes : cap) / 64), 64, 0, (hipStream_t)st>>>(
        reinterpret_cast<const unsigned char*>(e), reinterpret_cast<unsigned char*>(x), reinterpret_cast<unsigned char*>(y), sign, n);
    return check_launch("ecn mulgen_get");
}
