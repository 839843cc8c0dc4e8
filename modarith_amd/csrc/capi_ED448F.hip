// modarith_amd/csrc/capi_ED448F.hip -- ecn_ed448_mul_get_batch: scalar multiplication fused with the affine export
// (csrc/ed28l.h: the ladder form), the call pattern ecnXXXmul + ecnXXXget of the reference's signature code (ed448.c:182-184).
#include "../../include/modarith_amd.h"
#include "capi_common.h"
#include "generated/params_X448.h"
#include "kernels.h"
#include "ed28.h"
#include "ed28l_k.h"

namespace ma {

// round 5, the ladder form (csrc/ed28l.h): one scalar multiplication per lane on the birationally equivalent Montgomery curve, no table,
// no LDS; (u, w) of the point come from the shared inversion in front, the Edwards (X : Y : Z) of the result go to the shared inversion
// behind (csrc/edlad_k.h).  Rounds 2-4 walked 150 signed 3-bit windows over a 4-entry table in a workspace slab (ed28.h
// ed448_mul_get_one: 2.0e7/s); the double multiplication (capi_ED448F2.hip) still does.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_ed448_lad(const unsigned char* e, size_t first, Ed28lWs ws) {
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= ws.m) return;
    using L = Ed28Lad;
    uint32_t x2[16], z2[16], x3[16], z3[16];
    bool e_odd;
    {
        spint ew[7];
        load_be_record<P_X448>(e, first + t, ew);
        e_odd = (ew[0] & 1) != 0;
        uint32_t u[16];
        ws.load_u(t, u);
        L::ladder(ew, u, x2, z2, x3, z3);
    }
    const uint32_t fl = ws.flags[t];
    Ed28::Ext R;
    L::recover([&](uint32_t* o) { ws.load_u(t, o); }, [&](uint32_t* o) { (void)ws.load_w(t, o); }, fl, e_odd, x2, z2, x3, z3, R, false);
    ws.store_xyz(t, R.X, R.Y, R.Z);
}

}  // namespace ma

using namespace ma;

extern "C" size_t ecn_ed448_mul_get_workspace_bytes(size_t n) { return ed28l_workspace_bytes(n); }

extern "C" int ecn_ed448_mul_get_batch(const char* e, const ma_spint* P, char* x, char* y, int* sign, size_t n, size_t ld,
                                       void* workspace, size_t workspace_bytes, void* st) {
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 7u) {
        set_error("ecn mul_get: byte records must be 8-byte aligned");
        return (int)hipErrorInvalidValue;
    }
    hipStream_t s = (hipStream_t)st;
    EdLadScratch ws(workspace, workspace_bytes, ed28l_workspace_bytes(n), 8, s);
    if (!ws.p) {
        set_error(std::string("ecn mul_get: no usable workspace -- " + std::string(ws.why) + " (pass ecn_ed448_mul_get_workspace_bytes(n) bytes; the library's own scratch pool is not available while the stream is being captured)"));
        return (int)hipErrorInvalidValue;
    }
    const unsigned char* eb = reinterpret_cast<const unsigned char*>(e);
    edlad_pipeline<LadT448, 1>(P, ld, reinterpret_cast<unsigned char*>(x), reinterpret_cast<unsigned char*>(y), sign, n, ws.p, s,
                               [&](size_t first, size_t m, const Ed28lWs& w) { k_ed448_lad<<<(unsigned)((m + 63) / 64), 64, 0, s>>>(eb, first, w); });
    return check_launch("ecn mul_get (ladder form)");
}
