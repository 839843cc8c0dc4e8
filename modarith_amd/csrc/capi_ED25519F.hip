// modarith_amd/csrc/capi_ED25519F.hip -- ecn_ed25519_mul_get_batch: scalar multiplication fused with the affine export
// (csrc/ed26.h), the call pattern ecnXXXmul + ecnXXXget of the reference's signature code (ed448.c:182-184).
#include "../../include/modarith_amd.h"
#include "capi_common.h"
#include "generated/curve_ED25519.h"
#include "kernels.h"
#include "ed26.h"

namespace ma {

// one scalar multiplication per lane, one wave per workgroup; the point and its 4-entry table live in registers, the recoded scalar
// in LDS (ed26.h W25519_3Lds: one byte per window, written before the point is loaded); a lane's element index is the
// wave-uniform base + lane, formed where it is used
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_ed25519_mul_get(const unsigned char* e, const spint* Pb, unsigned char* xb, unsigned char* yb, int* sign, size_t n, size_t ld) {
    using P = P_X25519;
    __shared__ unsigned char digs[86 * 64];
    unsigned char* col = digs + threadIdx.x;
    __shared__ uint64_t parked[24 * 64];                     // entries 3P, 4P of the window table (ed26.h Park24Lds)
    Park24Lds park{parked + threadIdx.x};
    for (size_t base = (size_t)blockIdx.x * 64; base < n; base += (size_t)gridDim.x * 64) {
        auto t = [&]() {
            unsigned l = threadIdx.x;
            asm volatile("" : "+v"(l));
            return base + l;
        };
        if (t() >= n) continue;
        {
            spint ew[4];
            load_be_record<P>(e, t(), ew);
            W25519_3Lds::fill(ew, col);
        }
        spint X[5], Y[5], Z[5], xw[4], yw[4];
        static_for<0, 5>([&](auto I) {
            X[I] = Pb[(size_t)I * ld + t()];
            Y[I] = Pb[(size_t)(5 + I) * ld + t()];
            Z[I] = Pb[(size_t)(10 + I) * ld + t()];
        });
        W25519_3Lds dig{col};
        ed25519_mul_get_dig<C_ED25519>(dig, park, X, Y, Z, xw, yw);
        if (xb) store_be_record<P>(xb, t(), xw);
        if (yb) store_be_record<P>(yb, t(), yw);
        if (sign) sign[t()] = !yb ? (int)(yw[0] & 1) : (!xb ? (int)(xw[0] & 1) : 0);
    }
}

}  // namespace ma

using namespace ma;

extern "C" size_t ecn_ed25519_mul_get_workspace_bytes(size_t) { return 0; }      // the table lives in registers

extern "C" int ecn_ed25519_mul_get_batch(const char* e, const ma_spint* P, char* x, char* y, int* sign, size_t n, size_t ld,
                                         void* /*workspace*/, size_t /*workspace_bytes*/, void* st) {
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 7u) {
        set_error("ecn mul_get: byte records must be 8-byte aligned");
        return (int)hipErrorInvalidValue;
    }
    // resident grid: 2 waves on each of the 1024 SIMDs, grid-stride over the batch
    const size_t lanes = (n + 63) / 64 * 64;
    const size_t cap = (size_t)2 * 1024 * 64;
    k_ed25519_mul_get<<<(unsigned)((lanes < cap ? lanes : cap) / 64), 64, 0, (hipStream_t)st>>>(
        reinterpret_cast<const unsigned char*>(e), P, reinterpret_cast<unsigned char*>(x), reinterpret_cast<unsigned char*>(y), sign, n, ld);
    return check_launch("ecn mul_get");
}
