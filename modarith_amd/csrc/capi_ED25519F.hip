// modarith_amd/csrc/capi_ED25519F.hip -- ecn_ed25519_mul_get_batch: scalar multiplication fused with the affine export
// (csrc/ed26.h), the call pattern ecnXXXmul + ecnXXXget of the reference's signature code (ed448.c:182-184).
#include "../../include/modarith_amd.h"
#include <string.h>
#include "capi_common.h"
#include "generated/curve_ED25519.h"
#include "kernels.h"
#include "ed26.h"
#include "ed26l_k.h"

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

// round 5, the ladder form (csrc/ed26l.h): one scalar multiplication per lane on the Montgomery curve, no table, no LDS; (u, w) of the
// point come from the shared inversion in front, the Edwards (X : Y : Z) of the result go to the shared inversion behind
// 163 VGPRs, no scratch: three waves per SIMD (a 128-register build for four waves measured the same rate, 1.128e8 mul_get/s, and
// spilled 39 registers in the recovery; two waves -- enforced from outside through an LDS claim -- lost 12 %: profiles/r05_lad_ab.log)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_ed25519_lad(const unsigned char* e, size_t first, Ed26lWs ws) {
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= ws.m) return;
    using L = Ed26Lad<C_ED25519>;
    uint32_t u[10], w[10], x2[10], z2[10], x3[10], z3[10];
    bool e_odd;
    {
        spint ew[4];
        load_be_record<P_X25519>(e, first + t, ew);
        e_odd = (ew[0] & 1) != 0;
        ed26l_load_u(ws, t, u);
        L::ladder(ew, u, x2, z2, x3, z3);
    }
    ed26l_load_u(ws, t, u);                     // (again: cheaper than ten registers across the ladder)
    const uint32_t fl = ed26l_load_w(ws, t, w);
    Ed26<C_ED25519>::Ext R;
    L::recover<false>(u, w, fl, e_odd, x2, z2, x3, z3, R);
    ed26l_store_xyz(ws, t, R.X, R.Y, R.Z);
}

}  // namespace ma

using namespace ma;

// MA_ED25519_FUSED=window: the round-2..4 kernel (3-bit windows, table in registers, one inversion pair per lane) for every batch
static bool ed25519_fused_window() {
    static bool v = [] { const char* s = getenv("MA_ED25519_FUSED"); return s && strcmp(s, "window") == 0; }();
    return v;
}
constexpr size_t ED25519_LAD_MIN = 4096;         // below this the shared inversions have nothing to share

extern "C" size_t ecn_ed25519_mul_get_workspace_bytes(size_t n) { return n >= ED25519_LAD_MIN ? ed26l_workspace_bytes(n) : 0; }

extern "C" int ecn_ed25519_mul_get_batch(const char* e, const ma_spint* P, char* x, char* y, int* sign, size_t n, size_t ld,
                                         void* workspace, size_t workspace_bytes, void* st) {
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 7u) {
        set_error("ecn mul_get: byte records must be 8-byte aligned");
        return (int)hipErrorInvalidValue;
    }
    hipStream_t s = (hipStream_t)st;
    if (n >= ED25519_LAD_MIN && !ed25519_fused_window()) {
        // the ladder form: the caller's workspace, or stream-ordered scratch of the library's own pool when none (or too little) was
        // passed -- ecn_ed25519_mul_get_workspace_bytes returned 0 up to round 4, and callers of that contract pass NULL
        const size_t need = ed26l_workspace_bytes(n);
        void* ws = (workspace && workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 7u) == 0) ? workspace : nullptr;
        void* own = nullptr;
        if (!ws) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (s == nullptr || (hipStreamIsCapturing(s, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone)) ws = own = scratch_alloc(need, s);
            else (void)hipGetLastError();
        }
        if (ws) {
            const unsigned char* eb = reinterpret_cast<const unsigned char*>(e);
            ed26l_pipeline<C_ED25519, 1>(P, ld, reinterpret_cast<unsigned char*>(x), reinterpret_cast<unsigned char*>(y), sign, n, ws, s,
                                         [&](size_t first, size_t m, const Ed26lWs& w) { k_ed25519_lad<<<(unsigned)((m + 63) / 64), 64, 0, s>>>(eb, first, w); });
            if (own) scratch_free(own, s);
            return check_launch("ecn mul_get (ladder form)");
        }
    }
    // the window form.  Resident grid: 2 waves on each of the 1024 SIMDs, grid-stride over the batch
    const size_t lanes = (n + 63) / 64 * 64;
    const size_t cap = (size_t)2 * 1024 * 64;
    k_ed25519_mul_get<<<(unsigned)((lanes < cap ? lanes : cap) / 64), 64, 0, (hipStream_t)st>>>(
        reinterpret_cast<const unsigned char*>(e), P, reinterpret_cast<unsigned char*>(x), reinterpret_cast<unsigned char*>(y), sign, n, ld);
    return check_launch("ecn mul_get");
}
