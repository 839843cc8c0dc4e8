// modarith_amd/csrc/capi_ED25519F2.hip -- ecn_ed25519_mul2_get_batch: double multiplication e*P + f*Q fused with the
// affine export (csrc/ed26.h), the verification pattern ecnXXXmul2 + ecnXXXget of the reference's signature code
// (ed448.c:305, nist256.c:251-254).
#include "../../include/modarith_amd.h"
#include "capi_common.h"
#include "generated/curve_ED25519.h"
#include "kernels.h"
#include "ed26.h"

namespace ma {

// P's table entries in registers, Q's parked in LDS (ed26.h Park24Lds), both recoded scalars in LDS (four 2-bit windows per byte),
// element index formed at use: see capi_ED25519F.hip
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_ed25519_mul2_get(const unsigned char* e, const spint* Pb, const unsigned char* f, const spint* Qb, unsigned char* xb, unsigned char* yb,
                        int* sign, size_t n, size_t ld) {
    using P = P_X25519;
    __shared__ unsigned char digs[2 * 33 * 64];
    __shared__ uint64_t parked[24 * 64];
    unsigned char* ce = digs + threadIdx.x;
    unsigned char* cf = ce + 33 * 64;
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
            W25519_2Lds::fill(ew, ce);
            load_be_record<P>(f, t(), ew);
            W25519_2Lds::fill(ew, cf);
        }
        spint PX[5], PY[5], PZ[5], QX[5], QY[5], QZ[5], xw[4], yw[4];
        static_for<0, 5>([&](auto I) {
            PX[I] = Pb[(size_t)I * ld + t()];
            PY[I] = Pb[(size_t)(5 + I) * ld + t()];
            PZ[I] = Pb[(size_t)(10 + I) * ld + t()];
            QX[I] = Qb[(size_t)I * ld + t()];
            QY[I] = Qb[(size_t)(5 + I) * ld + t()];
            QZ[I] = Qb[(size_t)(10 + I) * ld + t()];
        });
        W25519_2Lds de{ce}, df{cf};
        ed25519_mul2_get_dig<C_ED25519>(de, PX, PY, PZ, df, QX, QY, QZ, park, xw, yw);
        if (xb) store_be_record<P>(xb, t(), xw);
        if (yb) store_be_record<P>(yb, t(), yw);
        if (sign) sign[t()] = !yb ? (int)(yw[0] & 1) : (!xb ? (int)(xw[0] & 1) : 0);
    }
}

}  // namespace ma

using namespace ma;

extern "C" size_t ecn_ed25519_mul2_get_workspace_bytes(size_t) { return 0; }      // both tables live in registers

extern "C" int ecn_ed25519_mul2_get_batch(const char* e, const ma_spint* P, const char* f, const ma_spint* Q, char* x, char* y, int* sign,
                                          size_t n, size_t ld, void* /*workspace*/, size_t /*workspace_bytes*/, void* st) {
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(f) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 7u) {
        set_error("ecn mul2_get: byte records must be 8-byte aligned");
        return (int)hipErrorInvalidValue;
    }
    const size_t lanes = (n + 63) / 64 * 64;
    const size_t cap = (size_t)2 * 1024 * 64;
    k_ed25519_mul2_get<<<(unsigned)((lanes < cap ? lanes : cap) / 64), 64, 0, (hipStream_t)st>>>(
        reinterpret_cast<const unsigned char*>(e), P, reinterpret_cast<const unsigned char*>(f), Q, reinterpret_cast<unsigned char*>(x),
        reinterpret_cast<unsigned char*>(y), sign, n, ld);
    return check_launch("ecn mul2_get");
}
