// modarith_amd/csrc/capi_SECP256K1F.hip -- ecn_secp256k1_mul_get_batch: secp256k1 scalar multiplication fused with the affine
// export (csrc/wn26.h), the call pattern ecnXXXmul + ecnXXXget of the reference's ECDSA code (nist256.c:155-161, 219-222; curve.py:190-198 builds the same layer for secp256k1).
#include "../../include/modarith_amd.h"
#include "capi_common.h"
#include "generated/curve_SECP256K1.h"
#include "kernels.h"
#include "wn26.h"
#include "glv26.h"
#include "wn_export.h"

namespace ma {

constexpr size_t SECP256K1_ROW_SKEW = 32 + 4;   // words added to the row pitch of the table workspace (as in capi_ED448F.hip)

// one scalar multiplication per lane, one wave per workgroup, the scalar split by the curve's endomorphism (csrc/glv26.h: 128 doublings
// instead of 256 on the same table): window tables in the wave's slab of the workspace ([word][64 lanes]: every
// access one contiguous 512-byte row, row addresses formed at the access -- wn26.h WnTabSlab), recoded scalar in LDS (one byte per
// window, written before the point is loaded), element index = wave-uniform base + lane, formed where it is used
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_secp256k1_mul_get(const unsigned char* e, const spint* Pb, size_t ld, uint64_t* ws, WnExpWs ex) {
    const size_t n = ex.m;                                  // the records of this chunk; (X : Y : Z) of the result to the shared inversion (wn_export.h)
    using P = P_SECP256K1;
    using DIG = GlvLds;
    __shared__ unsigned char digs[DIG::COUNT * 64];
    const WnTabSlab T{ws + (size_t)blockIdx.x * (64 * (size_t)WN26_TABLE_WORDS), threadIdx.x};
    unsigned char* col = digs + threadIdx.x;
    for (size_t base = (size_t)blockIdx.x * 64; base < n; base += (size_t)gridDim.x * 64) {
        auto t = [&]() { return base + (size_t)(T.origin() - T.base); };
        if (t() >= n) continue;
        DIG dig;
        {
            spint ew[4];
            load_be_record<P>(e, t(), ew);
            dig.fill(ew, col);
        }
        spint X[5], Y[5], Z[5];
        static_for<0, 5>([&](auto I) {
            X[I] = Pb[(size_t)I * ld + t()];
            Y[I] = Pb[(size_t)(5 + I) * ld + t()];
            Z[I] = Pb[(size_t)(10 + I) * ld + t()];
        });
        Wn26<CvSecp256k1>::Pt R;
        secp256k1_glv_mul_acc(dig, X, Y, Z, T, R);
        ex.store<Fk26>(t(), R.X, R.Y, R.Z);
    }
}

}  // namespace ma

using namespace ma;

namespace {
// resident grid: 2 waves on each of the 1024 SIMDs, grid-stride over the batch
size_t fused_lanes(size_t n) {
    const size_t lanes = (n + 63) / 64 * 64, cap = (size_t)3 * 1024 * 64;
    return lanes < cap ? lanes : cap;
}
}  // namespace

static size_t slab_bytes(size_t n) { return (fused_lanes(n) + SECP256K1_ROW_SKEW) * WN26_TABLE_WORDS * sizeof(uint64_t); }
extern "C" size_t ecn_secp256k1_mul_get_workspace_bytes(size_t n) { return slab_bytes(n) + WnExpWs::bytes(n); }

extern "C" int ecn_secp256k1_mul_get_batch(const char* e, const ma_spint* P, char* x, char* y, int* sign, size_t n, size_t ld,
                                         void* workspace, size_t workspace_bytes, void* st) {
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 7u) {
        set_error("ecn mul_get: byte records must be 8-byte aligned");
        return (int)hipErrorInvalidValue;
    }
    if (workspace == nullptr || (reinterpret_cast<uintptr_t>(workspace) & 7u) || workspace_bytes < ecn_secp256k1_mul_get_workspace_bytes(n)) {
        set_error("ecn mul_get: workspace missing, not 8-byte aligned or too small (see ecn_secp256k1_mul_get_workspace_bytes)");
        return (int)hipErrorInvalidValue;
    }
    hipStream_t s = (hipStream_t)st;
    for (size_t first = 0; first < n; first += WNEXP_CHUNK) {
        const size_t m = n - first < WNEXP_CHUNK ? n - first : WNEXP_CHUNK;
        const WnExpWs ex(reinterpret_cast<char*>(workspace) + slab_bytes(n), m);
        k_secp256k1_mul_get<<<(unsigned)(fused_lanes(m) / 64), 64, 0, s>>>(reinterpret_cast<const unsigned char*>(e) + first * P_SECP256K1::NBYTES, P + first, ld, reinterpret_cast<uint64_t*>(workspace), ex);
        wn_export<Fk26, P_SECP256K1, 1>(ex, reinterpret_cast<unsigned char*>(x), reinterpret_cast<unsigned char*>(y), sign, first, s);
    }
    return check_launch("ecn mul_get");
}
