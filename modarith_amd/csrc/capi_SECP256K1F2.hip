// modarith_amd/csrc/capi_SECP256K1F2.hip -- ecn_secp256k1_mul2_get_batch: double multiplication e*P + f*Q fused with the
// affine export (csrc/wn26.h), the verification pattern ecnXXXmul2 + ecnXXXget of the reference's ECDSA code
// (nist256.c:251-256).  A result at infinity leaves as x = 0, y = 1 (what ecnXXXget gives; the caller's ecnXXXisinf test
// becomes x == 0 && y == 1, no point of the curve has x = 0 ... y = 1 since b = 7 is not 1).
#include "../../include/modarith_amd.h"
#include "capi_common.h"
#include "generated/curve_SECP256K1.h"
#include "kernels.h"
#include "wn26.h"
#include "glv26.h"
#include "wn_export.h"

namespace ma {

constexpr size_t SECP256K1_ROW_SKEW2 = 32 + 4;

// both scalars split by the endomorphism (csrc/glv26.h), the two tables of eight in the wave's slab (a double slot), the four recoded
// half scalars in LDS, element index formed at use: see the mul_get unit
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_secp256k1_mul2_get(const unsigned char* e, const spint* Pb, const unsigned char* f, const spint* Qb, size_t ld, uint64_t* ws, WnExpWs ex) {
    const size_t n = ex.m;                                  // the records of this chunk; (X : Y : Z) of the result to the shared inversion (wn_export.h)
    using P = P_SECP256K1;
    using DIG = GlvLds;
    __shared__ unsigned char digs[2 * DIG::COUNT * 64];
    const WnTabSlab T{ws + (size_t)blockIdx.x * (64 * (size_t)GLV2_TABLE_WORDS), threadIdx.x};
    unsigned char* ce = digs + threadIdx.x;
    unsigned char* cf = ce + DIG::COUNT * 64;
    for (size_t base = (size_t)blockIdx.x * 64; base < n; base += (size_t)gridDim.x * 64) {
        auto t = [&]() { return base + (size_t)(T.origin() - T.base); };
        if (t() >= n) continue;
        DIG de, df;
        {
            spint ew[4];
            load_be_record<P>(e, t(), ew);
            de.fill(ew, ce);
            load_be_record<P>(f, t(), ew);
            df.fill(ew, cf);
        }
        auto point = [&](const spint* B) {                   // the 3 x 5 limbs of record t() of a point batch, fetched when its table is built
            return [&, B](spint* X, spint* Y, spint* Z) {
                static_for<0, 5>([&](auto I) {
                    X[I] = B[(size_t)I * ld + t()];
                    Y[I] = B[(size_t)(5 + I) * ld + t()];
                    Z[I] = B[(size_t)(10 + I) * ld + t()];
                });
            };
        };
        Wn26<CvSecp256k1>::Pt R;
        secp256k1_glv_mul2_acc_ld(de, point(Pb), df, point(Qb), T, R);
        ex.store<Fk26>(t(), R.X, R.Y, R.Z);
    }
}

}  // namespace ma

using namespace ma;

namespace {
size_t fused_lanes(size_t n) {
    const size_t lanes = (n + 63) / 64 * 64, cap = (size_t)3 * 1024 * 64;
    return lanes < cap ? lanes : cap;
}
}  // namespace

static size_t slab_bytes(size_t n) { return (fused_lanes(n) + SECP256K1_ROW_SKEW2) * GLV2_TABLE_WORDS * sizeof(uint64_t); }
extern "C" size_t ecn_secp256k1_mul2_get_workspace_bytes(size_t n) { return slab_bytes(n) + WnExpWs::bytes(n); }

extern "C" int ecn_secp256k1_mul2_get_batch(const char* e, const ma_spint* P, const char* f, const ma_spint* Q, char* x, char* y, int* sign,
                                          size_t n, size_t ld, void* workspace, size_t workspace_bytes, void* st) {
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(f) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 7u) {
        set_error("ecn mul2_get: byte records must be 8-byte aligned");
        return (int)hipErrorInvalidValue;
    }
    if (workspace == nullptr || (reinterpret_cast<uintptr_t>(workspace) & 7u) || workspace_bytes < ecn_secp256k1_mul2_get_workspace_bytes(n)) {
        set_error("ecn mul2_get: workspace missing, not 8-byte aligned or too small (see ecn_secp256k1_mul2_get_workspace_bytes)");
        return (int)hipErrorInvalidValue;
    }
    hipStream_t s = (hipStream_t)st;
    for (size_t first = 0; first < n; first += WNEXP_CHUNK) {
        const size_t m = n - first < WNEXP_CHUNK ? n - first : WNEXP_CHUNK;
        const WnExpWs ex(reinterpret_cast<char*>(workspace) + slab_bytes(n), m);
        k_secp256k1_mul2_get<<<(unsigned)(fused_lanes(m) / 64), 64, 0, s>>>(reinterpret_cast<const unsigned char*>(e) + first * P_SECP256K1::NBYTES, P + first, reinterpret_cast<const unsigned char*>(f) + first * P_SECP256K1::NBYTES, Q + first, ld, reinterpret_cast<uint64_t*>(workspace), ex);
        wn_export<Fk26, P_SECP256K1, 2>(ex, reinterpret_cast<unsigned char*>(x), reinterpret_cast<unsigned char*>(y), sign, first, s);
    }
    return check_launch("ecn mul2_get");
}
