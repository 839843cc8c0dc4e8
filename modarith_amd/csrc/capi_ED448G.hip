// modarith_amd/csrc/capi_ED448G.hip -- ecn_ed448_mulgen_get_batch: generator multiplication fused with the affine export
// (csrc/ed28.h ed448_mulgen_get_one), the call sequence ecnXXXgen + ecnXXXmul + ecnXXXget that opens EdDSA key generation
// and signing in the reference (ed448.c:167-184, 196-199).  Fixed-base table: generated/comb_ED448.h.
#include "../../include/modarith_amd.h"
#include "capi_common.h"
#include "generated/params_X448.h"
#include "generated/comb_ED448.h"
#include "kernels.h"
#include "ed28.h"

namespace ma {

// COMB_ED448_WINDOWS windows of COMB_ED448_W bits x 2^(W-1) multiples x coordinates x limbs, the same for every lane: constant address space, wave-uniform indices
__constant__ int32_t comb_ed448[] = { COMB_ED448_VALUES };
struct CombED448 {
    static constexpr int W = COMB_ED448_W, NW = COMB_ED448_WINDOWS;
    static __device__ __forceinline__ int32_t get(int idx) { return comb_ed448[idx]; }
};

// the first of the two results of a lane waits here (64 words per lane, [word][lane]: conflict-free; the second result's X, Y join it across the shared inversion) while the second scalar runs
struct LdsPark {
    uint32_t* base;
    __device__ __forceinline__ void put(int k, uint32_t v) { base[k * 64] = v; }
    __device__ __forceinline__ uint32_t get(int k) const { return base[k * 64]; }
};

// two scalars per lane (elements t and t + lanes of a 2 * lanes stride) share one inversion
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_ed448_mulgen_get(const unsigned char* e, unsigned char* xb, unsigned char* yb, int* sign, size_t n) {
    using P = P_X448;
    __shared__ uint32_t lds[64 * 64];
    LdsPark park{lds + threadIdx.x};
    const size_t lanes = (size_t)gridDim.x * blockDim.x;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += 2 * lanes) {
        spint xw[2][7], yw[2][7];
        ed448_mulgen_get_two<CombED448>(
            [&](int g, spint* ew) { const size_t tg = t + (size_t)g * lanes; load_be_record<P>(e, tg < n ? tg : t, ew); }, park, xw, yw);
        static_for<0, 2>([&](auto GI) {
            const size_t tg = t + (size_t)GI * lanes;
            if (tg < n) {
                if (xb) store_be_record<P>(xb, tg, xw[GI]);
                if (yb) store_be_record<P>(yb, tg, yw[GI]);
                if (sign) sign[tg] = !yb ? (int)(yw[GI][0] & 1) : (!xb ? (int)(xw[GI][0] & 1) : 0);
            }
        });
    }
}

// rfc7748() on the base point u = 5 (x448_base_one): little-endian 56-byte records as rfc7748_X448_batch takes them
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_x448_base(const uint64_t* bk, uint64_t* bv, size_t n) {
    __shared__ uint32_t lds[64 * 64];
    LdsPark park{lds + threadIdx.x};
    const size_t lanes = (size_t)gridDim.x * blockDim.x;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += 2 * lanes) {
        uint64_t ow[2][7];
        x448_base_two<CombED448>(
            [&](int g, uint64_t* kw) {
                const size_t tg = t + (size_t)g * lanes, ts = tg < n ? tg : t;
                static_for<0, 7>([&](auto K) { kw[K] = bk[ts * 7 + K]; });
            }, park, ow);
        static_for<0, 2>([&](auto GI) {
            const size_t tg = t + (size_t)GI * lanes;
            if (tg < n) static_for<0, 7>([&](auto K) { bv[tg * 7 + K] = ow[GI][K]; });
        });
    }
}

// e*G + f*Q and its affine export (ED448_VERIFY, ed448.c:290-310): the per-lane table of Q in the workspace as for mul_get, the
// generator part through the constant table above
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_ed448_mulgen2_get(const unsigned char* e, const unsigned char* f, const spint* Qb, unsigned char* xb, unsigned char* yb, int* sign,
                         size_t n, size_t ld, uint64_t* ws) {
    using P = P_X448;
    __shared__ unsigned char digs[150 * 64];                 // f's windows (ed28.h Win3Lds); Q's table in the wave's slab: capi_ED448F.hip
    const TabSlab T{ws + (size_t)blockIdx.x * (64 * (size_t)ED448_TABLE_WORDS), threadIdx.x};
    unsigned char* col = digs + threadIdx.x;
    for (size_t base = (size_t)blockIdx.x * 64; base < n; base += (size_t)gridDim.x * 64) {
        auto t = [&]() { return base + (size_t)(T.origin() - T.base); };
        if (t() >= n) continue;
        {
            spint fw[7];
            load_be_record<P>(f, t(), fw);
            Win3Lds::fill(fw, col);
        }
        spint ew[7], X[8], Y[8], Z[8], xw[7], yw[7];
        static_for<0, 8>([&](auto I) {
            X[I] = Qb[(size_t)I * ld + t()];
            Y[I] = Qb[(size_t)(8 + I) * ld + t()];
            Z[I] = Qb[(size_t)(16 + I) * ld + t()];
        });
        Win3Lds dig{col};
        Ed28::Ext R;
        ed448_mul_acc<true>(dig, X, Y, Z, T, R);             // f*Q, leaving with its T coordinate
        load_be_record<P>(e, t(), ew);                       // e is not needed (nor held) before this point
        ed448_mulgen_acc<CombED448, false>(ew, R);           // += e*G through the fixed-base table
        {
            using F = Fe28;
            uint32_t zi[16], ax[16], ay[16];
            F::invert(R.Z, zi);
            F::mul_k(R.X, zi, ax);
            F::mul_k(R.Y, zi, ay);
            F::to_words(ax, xw);
            F::to_words(ay, yw);
        }
        if (xb) store_be_record<P>(xb, t(), xw);
        if (yb) store_be_record<P>(yb, t(), yw);
        if (sign) sign[t()] = !yb ? (int)(yw[0] & 1) : (!xb ? (int)(xw[0] & 1) : 0);
    }
}

}  // namespace ma

using namespace ma;

namespace {
size_t fused2_lanes(size_t n) {
    const size_t lanes = (n + 63) / 64 * 64, cap = (size_t)2 * 1024 * 64;
    return lanes < cap ? lanes : cap;
}
}  // namespace

extern "C" size_t ecn_ed448_mulgen2_get_workspace_bytes(size_t n) { return (fused2_lanes(n) + 36) * ED448_TABLE_WORDS * sizeof(uint64_t); }

extern "C" int ecn_ed448_mulgen2_get_batch(const char* e, const char* f, const ma_spint* Q, char* x, char* y, int* sign, size_t n, size_t ld,
                                           void* workspace, size_t workspace_bytes, void* st) {
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(f) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 7u) {
        set_error("ecn mulgen2_get: byte records must be 8-byte aligned");
        return (int)hipErrorInvalidValue;
    }
    const size_t lanes = fused2_lanes(n);
    if (workspace == nullptr || workspace_bytes < (lanes + 36) * ED448_TABLE_WORDS * sizeof(uint64_t)) {
        set_error("ecn mulgen2_get: workspace too small (see ecn_ed448_mulgen2_get_workspace_bytes)");
        return (int)hipErrorInvalidValue;
    }
    k_ed448_mulgen2_get<<<(unsigned)(lanes / 64), 64, 0, (hipStream_t)st>>>(
        reinterpret_cast<const unsigned char*>(e), reinterpret_cast<const unsigned char*>(f), Q, reinterpret_cast<unsigned char*>(x),
        reinterpret_cast<unsigned char*>(y), sign, n, ld, reinterpret_cast<uint64_t*>(workspace));
    return check_launch("ecn mulgen2_get");
}

extern "C" int ecn_ed448_mulgen_get_batch(const char* e, char* x, char* y, int* sign, size_t n, void* st) {
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 7u) {
        set_error("ecn mulgen_get: byte records must be 8-byte aligned");
        return (int)hipErrorInvalidValue;
    }
    const size_t lanes = ((n + 1) / 2 + 63) / 64 * 64, cap = (size_t)2 * 1024 * 64;       // two scalars per lane; two waves on each of the 1024 SIMDs
    k_ed448_mulgen_get<<<(unsigned)((lanes < cap ? lanes : cap) / 64), 64, 0, (hipStream_t)st>>>(
        reinterpret_cast<const unsigned char*>(e), reinterpret_cast<unsigned char*>(x), reinterpret_cast<unsigned char*>(y), sign, n);
    return check_launch("ecn mulgen_get");
}

// bv = [bk](5): rfc7748(bk, base, bv) for a batch of private keys, on the fixed-base table of ED448 (4-isogenous to curve448)
extern "C" int rfc7748_X448_base_batch(const char* bk, char* bv, size_t n, void* st) {
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(bk) | reinterpret_cast<uintptr_t>(bv)) & 7u) {
        set_error("rfc7748 base: byte records must be 8-byte aligned");
        return (int)hipErrorInvalidValue;
    }
    const size_t lanes = ((n + 1) / 2 + 63) / 64 * 64, cap = (size_t)2 * 1024 * 64;
    k_x448_base<<<(unsigned)((lanes < cap ? lanes : cap) / 64), 64, 0, (hipStream_t)st>>>(
        reinterpret_cast<const uint64_t*>(bk), reinterpret_cast<uint64_t*>(bv), n);
    return check_launch("rfc7748 base");
}
