// modarith_amd/csrc/capi_NIST256F2.hip -- ecn_nist256_mul2_get_batch: double multiplication e*P + f*Q fused with the
// affine export, the verification pattern ecnXXXmul2 + ecnXXXget of the reference's ECDSA code (nist256.c:251-256).  A result at
// infinity leaves as x = 0, y = 1 (what ecnXXXget gives; the caller's ecnXXXisinf test becomes x == 0 && y == 1, no point of the curve
// has x = 0 ... y = 1 since b is not 1).  Round 5 (csrc/wj26.h mul2_acc_aff): the two window tables of every record brought to Z = 1
// (csrc/wn_affine.h), a Jacobian accumulator with mixed additions, the additions where it meets +- a table point detected and redone
// with the complete formula (variable time: the inputs of a verification are public, the reference's own mul2 branches on them).
#include "../../include/modarith_amd.h"
#include "capi_common.h"
#include "generated/curve_NIST256.h"
#include "kernels.h"
#include "wn26.h"
#include "wj26.h"
#include "wn_export.h"

namespace ma {

// the multiples 1..8 of P (entries 0..7) and of Q (8..15) of every record of the chunk (Jacobian), one record per lane
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_nist256_tables(const spint* Pb, const spint* Qb, size_t ld, WnAffWs ws) {
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= ws.m) return;
    auto point = [&](const spint* B) {
        return [&, B](spint* X, spint* Y, spint* Z) {
            static_for<0, 5>([&](auto I) {
                X[I] = B[(size_t)I * ld + t];
                Y[I] = B[(size_t)(5 + I) * ld + t];
                Z[I] = B[(size_t)(10 + I) * ld + t];
            });
        };
    };
    Wj26::table_of(point(Pb), ws, t, 0);
    Wj26::table_of(point(Qb), ws, t, 8);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_nist256_mul2_get(const unsigned char* e, const unsigned char* f, WnAffWs ws, WnExpWs ex) {
    using P = P_NIST256;
    using DIG = WnLds<4, 260>;
    __shared__ unsigned char digs[2 * DIG::ROWS * 64];
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= ws.m) return;
    unsigned char* ce = digs + threadIdx.x;
    unsigned char* cf = ce + DIG::ROWS * 64;
    {
        spint ew[4];
        load_be_record<P>(e, t, ew);
        DIG::fill(ew, ce);
        load_be_record<P>(f, t, ew);
        DIG::fill(ew, cf);
    }
    DIG de{ce}, df{cf};
    Wj26::Pt R;
    Wj26::mul2_acc_aff(de, df, ws, t, R);
    ex.store<Fm26>(t, R.X, R.Y, R.Z);
}

}  // namespace ma

using namespace ma;

extern "C" size_t ecn_nist256_mul2_get_workspace_bytes(size_t n) { return WnAffWs::bytes(n, 16) + WnExpWs::bytes(n); }

extern "C" int ecn_nist256_mul2_get_batch(const char* e, const ma_spint* P, const char* f, const ma_spint* Q, char* x, char* y, int* sign,
                                          size_t n, size_t ld, void* workspace, size_t workspace_bytes, void* st) {
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(f) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 7u) {
        set_error("ecn mul2_get: byte records must be 8-byte aligned");
        return (int)hipErrorInvalidValue;
    }
    if (workspace == nullptr || (reinterpret_cast<uintptr_t>(workspace) & 7u) || workspace_bytes < ecn_nist256_mul2_get_workspace_bytes(n)) {
        set_error("ecn mul2_get: workspace missing, not 8-byte aligned or too small (see ecn_nist256_mul2_get_workspace_bytes)");
        return (int)hipErrorInvalidValue;
    }
    hipStream_t s = (hipStream_t)st;
    char* wsb = reinterpret_cast<char*>(workspace);
    for (size_t first = 0; first < n; first += WNAFF_CHUNK) {
        const size_t m = n - first < WNAFF_CHUNK ? n - first : WNAFF_CHUNK;
        const WnAffWs aw(wsb, m, 16);
        const WnExpWs ex(wsb + WnAffWs::bytes(n, 16), m);
        const unsigned g = (unsigned)((m + 63) / 64);
        k_nist256_tables<<<g, 64, 0, s>>>(P + first, Q + first, ld, aw);
        wn_table_affine<Fm26, true, 2>(aw, s);
        k_nist256_mul2_get<<<g, 64, 0, s>>>(reinterpret_cast<const unsigned char*>(e) + first * P_NIST256::NBYTES,
                                            reinterpret_cast<const unsigned char*>(f) + first * P_NIST256::NBYTES, aw, ex);
        wn_export<Fm26, P_NIST256, 2>(ex, reinterpret_cast<unsigned char*>(x), reinterpret_cast<unsigned char*>(y), sign, first, s);
    }
    return check_launch("ecn mul2_get");
}
