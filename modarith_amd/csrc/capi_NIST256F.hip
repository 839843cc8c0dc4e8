// modarith_amd/csrc/capi_NIST256F.hip -- ecn_nist256_mul_get_batch: P-256 scalar multiplication fused with the affine
// export, the call pattern ecnXXXmul + ecnXXXget of the reference's ECDSA code (nist256.c:155-161, 219-222).  Round 5: Jacobian
// coordinates with the exceptional cases decided by the scalar (csrc/wj26.h), the window table of every record brought to Z = 1
// (csrc/wn_affine.h) so that the window loop runs mixed additions, the export through a shared inversion (csrc/wn_export.h).
#include "../../include/modarith_amd.h"
#include "capi_common.h"
#include "generated/curve_NIST256.h"
#include "kernels.h"
#include "wn26.h"
#include "wj26.h"
#include "wn_export.h"

namespace ma {

// the multiples P .. 8P of every record of the chunk (Jacobian), one record per lane
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_nist256_table(const spint* Pb, size_t ld, WnAffWs ws) {
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= ws.m) return;
    Wj26::table_of([&](spint* X, spint* Y, spint* Z) {
        static_for<0, 5>([&](auto I) {
            X[I] = Pb[(size_t)I * ld + t];
            Y[I] = Pb[(size_t)(5 + I) * ld + t];
            Z[I] = Pb[(size_t)(10 + I) * ld + t];
        });
    }, ws, t);
}
// the window loop on the affine table: one record per lane, the recoded scalar in LDS (one byte per window), (X : Y : Z) of the result to
// the shared inversion
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_nist256_mul_get(const unsigned char* e, WnAffWs ws, WnExpWs ex) {
    using P = P_NIST256;
    using DIG = WnLds<4, 260>;
    __shared__ unsigned char digs[DIG::ROWS * 64];
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= ws.m) return;
    unsigned char* col = digs + threadIdx.x;
    {
        spint ew[4], kw[4];
        load_be_record<P>(e, t, ew);
        Wj26::reduce_scalar(ew, kw);
        DIG::fill(kw, col);
    }
    DIG dig{col};
    Wj26::Pt R;
    Wj26::mul_acc_aff(dig, ws, t, R);
    ex.store<Fm26>(t, R.X, R.Y, R.Z);
}

}  // namespace ma

using namespace ma;

extern "C" size_t ecn_nist256_mul_get_workspace_bytes(size_t n) { return WnAffWs::bytes(n) + WnExpWs::bytes(n); }

extern "C" int ecn_nist256_mul_get_batch(const char* e, const ma_spint* P, char* x, char* y, int* sign, size_t n, size_t ld,
                                         void* workspace, size_t workspace_bytes, void* st) {
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 7u) {
        set_error("ecn mul_get: byte records must be 8-byte aligned");
        return (int)hipErrorInvalidValue;
    }
    if (workspace == nullptr || (reinterpret_cast<uintptr_t>(workspace) & 7u) || workspace_bytes < ecn_nist256_mul_get_workspace_bytes(n)) {
        set_error("ecn mul_get: workspace missing, not 8-byte aligned or too small (see ecn_nist256_mul_get_workspace_bytes)");
        return (int)hipErrorInvalidValue;
    }
    hipStream_t s = (hipStream_t)st;
    const unsigned char* eb = reinterpret_cast<const unsigned char*>(e);
    char* wsb = reinterpret_cast<char*>(workspace);
    for (size_t first = 0; first < n; first += WNAFF_CHUNK) {
        const size_t m = n - first < WNAFF_CHUNK ? n - first : WNAFF_CHUNK;
        const WnAffWs aw(wsb, m);
        const WnExpWs ex(wsb + WnAffWs::bytes(n), m);
        const unsigned g = (unsigned)((m + 63) / 64);
        k_nist256_table<<<g, 64, 0, s>>>(P + first, ld, aw);
        wn_table_affine<Fm26, true, 1>(aw, s);
        k_nist256_mul_get<<<g, 64, 0, s>>>(eb + first * P_NIST256::NBYTES, aw, ex);
        wn_export<Fm26, P_NIST256, 1>(ex, reinterpret_cast<unsigned char*>(x), reinterpret_cast<unsigned char*>(y), sign, first, s);
    }
    return check_launch("ecn mul_get");
}
