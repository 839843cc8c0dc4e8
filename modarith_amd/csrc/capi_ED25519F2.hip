// modarith_amd/csrc/capi_ED25519F2.hip -- ecn_ed25519_mul2_get_batch: double multiplication e*P + f*Q fused with the
// affine export (csrc/ed26s.h: the Straus form), the verification pattern ecnXXXmul2 + ecnXXXget of the reference's signature code
// (ed448.c:305, nist256.c:251-254).
#include "../../include/modarith_amd.h"
#include "capi_common.h"
#include "generated/curve_ED25519.h"
#include "kernels.h"
#include "ed26.h"
#include "ed26s.h"
#include "ed26l_k.h"

namespace ma {

// round 5, the Straus form (csrc/ed26s.h): signed 4-bit windows of both scalars, the two 9-entry tables in the wave's slab of the
// workspace (one 128-byte line per entry and lane, read by index), the recoded scalars in LDS, the Edwards (X : Y : Z) of the sum to
// the shared inversion of ed26l_k.h.  Records first .. first + ws.m of the caller's arrays.
#ifndef MA_STRAUS_WAVES
#define MA_STRAUS_WAVES 2
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MA_STRAUS_WAVES, MA_STRAUS_WAVES)))
void k_ed25519_mul2_straus(const unsigned char* e, const spint* Pb, const unsigned char* f, const spint* Qb, size_t first, size_t ld,
                           uint64_t* slab, Ed26lWs ws) {
    using P = P_X25519;
    using S = Ed26Straus<C_ED25519>;
    __shared__ unsigned char digs[2 * 65 * 64];
    unsigned char* ce = digs + threadIdx.x;
    unsigned char* cf = ce + 65 * 64;
    StrausTabSlab tab{slab + ((size_t)blockIdx.x * 64 + threadIdx.x) * (18 * 16)};
    for (size_t base = (size_t)blockIdx.x * 64; base < ws.m; base += (size_t)gridDim.x * 64) {
        auto t = [&]() {
            unsigned l = threadIdx.x;
            asm volatile("" : "+v"(l));
            return base + l;
        };
        if (t() >= ws.m) continue;
        {
            spint w[4];
            load_be_record<P>(e, first + t(), w);
            W25519_4Lds::fill(w, ce);
            load_be_record<P>(f, first + t(), w);
            W25519_4Lds::fill(w, cf);
        }
#pragma unroll 1
        for (int which = 0; which < 2; which++) {
            const spint* B = which ? Qb : Pb;
            spint X[5], Y[5], Z[5];
            static_for<0, 5>([&](auto I) {
                X[I] = B[(size_t)I * ld + first + t()];
                Y[I] = B[(size_t)(5 + I) * ld + first + t()];
                Z[I] = B[(size_t)(10 + I) * ld + first + t()];
            });
            S::build(tab, which, X, Y, Z);
        }
        W25519_4Lds de{ce}, df{cf};
        S::Ext R;
        S::walk(de, df, tab, R);
        ws.store_xyz(t(), R.X, R.Y, R.Z);
    }
}

}  // namespace ma

using namespace ma;

static size_t straus_waves(size_t n) {
    const size_t w = (n + 63) / 64, cap = (size_t)MA_STRAUS_WAVES * 1024;
    return w < cap ? w : cap;
}

static size_t mul2_get_net_bytes(size_t n) {
    const size_t m = n < EDLAD_CHUNK ? n : EDLAD_CHUNK;
    return straus_waves(m) * STRAUS_SLAB_BYTES_PER_WAVE + ed26l_workspace_bytes(n);
}
// (+ 127: the table slab wants 128-byte alignment, EdLadScratch rounds the caller's pointer up)
extern "C" size_t ecn_ed25519_mul2_get_workspace_bytes(size_t n) { return mul2_get_net_bytes(n) + 127; }

extern "C" int ecn_ed25519_mul2_get_batch(const char* e, const ma_spint* P, const char* f, const ma_spint* Q, char* x, char* y, int* sign,
                                          size_t n, size_t ld, void* workspace, size_t workspace_bytes, void* st) {
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(f) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 7u) {
        set_error("ecn mul2_get: byte records must be 8-byte aligned");
        return (int)hipErrorInvalidValue;
    }
    hipStream_t s = (hipStream_t)st;
    EdLadScratch wsp(workspace, workspace_bytes, mul2_get_net_bytes(n), 128, s);
    if (!wsp.p) {
        set_error(std::string("ecn mul2_get: no usable workspace -- " + std::string(wsp.why) + " (pass ecn_ed25519_mul2_get_workspace_bytes(n) bytes; the library's own scratch pool is not available while the stream is being captured)"));
        return (int)hipErrorInvalidValue;
    }
    const unsigned char *eb = reinterpret_cast<const unsigned char*>(e), *fb = reinterpret_cast<const unsigned char*>(f);
    const size_t slab_bytes = straus_waves(n < EDLAD_CHUNK ? n : EDLAD_CHUNK) * STRAUS_SLAB_BYTES_PER_WAVE;
    for (size_t first = 0; first < n; first += EDLAD_CHUNK) {
        const size_t m = n - first < EDLAD_CHUNK ? n - first : EDLAD_CHUNK;
        Ed26lWs ws(reinterpret_cast<unsigned char*>(wsp.p) + slab_bytes, m);
        k_ed25519_mul2_straus<<<(unsigned)straus_waves(m), 64, 0, s>>>(eb, P, fb, Q, first, ld, reinterpret_cast<uint64_t*>(wsp.p), ws);
        edlad_export<LadT25519, 3>(ws, reinterpret_cast<unsigned char*>(x), reinterpret_cast<unsigned char*>(y), sign, first, s);
    }
    return check_launch("ecn mul2_get (Straus form)");
}

