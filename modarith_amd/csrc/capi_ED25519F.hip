// modarith_amd/csrc/capi_ED25519F.hip -- ecn_ed25519_mul_get_batch: scalar multiplication fused with the affine export
// (csrc/ed26l.h: the ladder form), the call pattern ecnXXXmul + ecnXXXget of the reference's signature code (ed448.c:182-184).
#include "../../include/modarith_amd.h"
#include "capi_common.h"
#include "generated/curve_ED25519.h"
#include "kernels.h"
#include "ed26.h"
#include "ed26l_k.h"

namespace ma {

// round 5, the ladder form (csrc/ed26l.h): one scalar multiplication per lane on the Montgomery curve, no table, no LDS; (u, w) of the
// point come from the shared inversion in front, the Edwards (X : Y : Z) of the result go to the shared inversion behind
// 163 VGPRs, no scratch: three waves per SIMD (a 128-register build for four waves measured the same rate, 1.128e8 mul_get/s, and
// spilled 39 registers in the recovery; two waves -- enforced from outside through an LDS claim -- lost 12 %: profiles/history/r05_lad_ab.log)
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
        ws.load_u(t, u);
        L::ladder(ew, u, x2, z2, x3, z3);
    }
    ws.load_u(t, u);                            // (again: cheaper than ten registers across the ladder)
    const uint32_t fl = ws.load_w(t, w);
    Ed26<C_ED25519>::Ext R;
    L::recover<false>(u, w, fl, e_odd, x2, z2, x3, z3, R);
    ws.store_xyz(t, R.X, R.Y, R.Z);
}

}  // namespace ma

using namespace ma;

extern "C" size_t ecn_ed25519_mul_get_workspace_bytes(size_t n) { return ed26l_workspace_bytes(n); }

extern "C" int ecn_ed25519_mul_get_batch(const char* e, const ma_spint* P, char* x, char* y, int* sign, size_t n, size_t ld,
                                         void* workspace, size_t workspace_bytes, void* st) {
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 7u) {
        set_error("ecn mul_get: byte records must be 8-byte aligned");
        return (int)hipErrorInvalidValue;
    }
    hipStream_t s = (hipStream_t)st;
    EdLadScratch ws(workspace, workspace_bytes, ed26l_workspace_bytes(n), 8, s);
    if (!ws.p) {
        set_error(std::string("ecn mul_get: no usable workspace -- " + std::string(ws.why) + " (pass ecn_ed25519_mul_get_workspace_bytes(n) bytes; the library's own scratch pool is not available while the stream is being captured)"));
        return (int)hipErrorInvalidValue;
    }
    const unsigned char* eb = reinterpret_cast<const unsigned char*>(e);
    edlad_pipeline<LadT25519, 1>(P, ld, reinterpret_cast<unsigned char*>(x), reinterpret_cast<unsigned char*>(y), sign, n, ws.p, s,
                                 [&](size_t first, size_t m, const Ed26lWs& w) { k_ed25519_lad<<<(unsigned)((m + 63) / 64), 64, 0, s>>>(eb, first, w); });
    return check_launch("ecn mul_get (ladder form)");
}
