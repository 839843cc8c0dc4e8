// modarith_amd/csrc/capi_NIST256G.hip -- ecn_nist256_mulgen_get_batch: generator multiplication fused with the affine export
// (csrc/wn26.h wn26_mulgen_get_one), the call sequence ecnXXXgen + ecnXXXmul + ecnXXXget that opens key generation and
// signing in the reference's ECDSA code (nist256.c:150-161, 214-222).  Fixed-base table: generated/comb_NIST256.h.
#include "../../include/modarith_amd.h"
#include "capi_common.h"
#include "generated/curve_NIST256.h"
#include "generated/comb_NIST256.h"
#include "kernels.h"
#include "wn26.h"

namespace ma {

// 65 windows x 8 multiples x (x, y) x 10 limbs, the same for every lane: constant address space, wave-uniform indices
__constant__ int32_t comb_nist256[65 * 8 * 2 * 10] = { COMB_NIST256_VALUES };
struct CombNIST256 {
    static __device__ __forceinline__ int32_t get(int idx) { return comb_nist256[idx]; }
};

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 4)))
void k_nist256_mulgen_get(const unsigned char* e, unsigned char* xb, unsigned char* yb, int* sign, size_t n) {
    using P = P_NIST256;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        spint ew[4], xw[4], yw[4];
        load_be_record<P>(e, t, ew);
        wn26_mulgen_get_one<CvNist256, CombNIST256>(ew, xw, yw);
        if (xb) store_be_record<P>(xb, t, xw);
        if (yb) store_be_record<P>(yb, t, yw);
        if (sign) sign[t] = !yb ? (int)(yw[0] & 1) : (!xb ? (int)(xw[0] & 1) : 0);
    }
}

}  // namespace ma

using namespace ma;

extern "C" int ecn_nist256_mulgen_get_batch(const char* e, char* x, char* y, int* sign, size_t n, void* st) {
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 7u) {
        set_error("ecn mulgen_get: byte records must be 8-byte aligned");
        return (int)hipErrorInvalidValue;
    }
    const size_t lanes = (n + 63) / 64 * 64, cap = (size_t)4 * 1024 * 64;       // at most 4 waves on each of the 1024 SIMDs
    k_nist256_mulgen_get<<<(unsigned)((lanes < cap ? lanes : cap) / 64), 64, 0, (hipStream_t)st>>>(
        reinterpret_cast<const unsigned char*>(e), reinterpret_cast<unsigned char*>(x), reinterpret_cast<unsigned char*>(y), sign, n);
    return check_launch("ecn mulgen_get");
}
