// modarith_amd/csrc/capi_ED448F.hip -- ecn_ed448_mul_get_batch: scalar multiplication fused with the affine export
// (csrc/ed28.h), the call pattern ecnXXXmul + ecnXXXget of the reference's signature code (ed448.c:182-184).
#include "../../include/modarith_amd.h"
#include "capi_common.h"
#include "generated/params_X448.h"
#include "kernels.h"
#include "ed28.h"

namespace ma {

constexpr size_t ED448_ROW_SKEW = 32 + 4;   // (round-3 layout: words added to the row pitch; the size the workspace query still reports)

// one scalar multiplication per lane, one wave per workgroup.  The wave's window tables sit in its slab of the workspace,
// [word][64 lanes] (every access one contiguous 512-byte row; row addresses formed at the access: ed28.h TabSlab); the recoded
// scalar sits in LDS, one byte per window (ed28.h Win3Lds), written before the point is loaded; a lane's element index is the
// wave-uniform base + lane, formed where it is used.  Round 3 kept the scalar words and ~84 row addresses in registers across
// the window loop: 791 spilled VGPRs, 350-480 scratch accesses per window.
#ifndef MA_ED448F_WAVES
#define MA_ED448F_WAVES 2
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MA_ED448F_WAVES, MA_ED448F_WAVES)))
void k_ed448_mul_get(const unsigned char* e, const spint* Pb, unsigned char* xb, unsigned char* yb, int* sign, size_t n, size_t ld,
                     uint64_t* ws) {
    using P = P_X448;
    __shared__ unsigned char digs[150 * 64];
    const TabSlab T{ws + (size_t)blockIdx.x * (64 * (size_t)ED448_TABLE_WORDS), threadIdx.x};
    unsigned char* col = digs + threadIdx.x;
    for (size_t base = (size_t)blockIdx.x * 64; base < n; base += (size_t)gridDim.x * 64) {
        auto t = [&]() { return base + (size_t)(T.origin() - T.base); };      // base + lane, as a fresh value at each use
        if (t() >= n) continue;
        {
            spint ew[7];
            load_be_record<P>(e, t(), ew);
            Win3Lds::fill(ew, col);
        }
        spint X[8], Y[8], Z[8], xw[7], yw[7];
        static_for<0, 8>([&](auto I) {
            X[I] = Pb[(size_t)I * ld + t()];
            Y[I] = Pb[(size_t)(8 + I) * ld + t()];
            Z[I] = Pb[(size_t)(16 + I) * ld + t()];
        });
        Win3Lds dig{col};
        ed448_mul_get_one(dig, X, Y, Z, T, xw, yw);
        if (xb) store_be_record<P>(xb, t(), xw);
        if (yb) store_be_record<P>(yb, t(), yw);
        if (sign) sign[t()] = !yb ? (int)(yw[0] & 1) : (!xb ? (int)(xw[0] & 1) : 0);
    }
}

}  // namespace ma

using namespace ma;

namespace {
// resident grid: 2 waves on each of the 1024 SIMDs, grid-stride over the batch
size_t fused_lanes(size_t n) {
    const size_t lanes = (n + 63) / 64 * 64, cap = (size_t)MA_ED448F_WAVES * 1024 * 64;
    return lanes < cap ? lanes : cap;
}
}  // namespace

extern "C" size_t ecn_ed448_mul_get_workspace_bytes(size_t n) { return (fused_lanes(n) + ED448_ROW_SKEW) * ED448_TABLE_WORDS * sizeof(uint64_t); }

extern "C" int ecn_ed448_mul_get_batch(const char* e, const ma_spint* P, char* x, char* y, int* sign, size_t n, size_t ld,
                                       void* workspace, size_t workspace_bytes, void* st) {
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 7u) {
        set_error("ecn mul_get: byte records must be 8-byte aligned");
        return (int)hipErrorInvalidValue;
    }
    const size_t lanes = fused_lanes(n);
    if (workspace == nullptr || workspace_bytes < (lanes + ED448_ROW_SKEW) * ED448_TABLE_WORDS * sizeof(uint64_t)) {
        set_error("ecn mul_get: workspace too small (see ecn_ed448_mul_get_workspace_bytes)");
        return (int)hipErrorInvalidValue;
    }
    k_ed448_mul_get<<<(unsigned)(lanes / 64), 64, 0, (hipStream_t)st>>>(
        reinterpret_cast<const unsigned char*>(e), P, reinterpret_cast<unsigned char*>(x), reinterpret_cast<unsigned char*>(y), sign, n, ld,
        reinterpret_cast<uint64_t*>(workspace));
    return check_launch("ecn mul_get");
}
