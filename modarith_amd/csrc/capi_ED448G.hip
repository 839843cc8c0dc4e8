// modarith_amd/csrc/capi_ED448G.hip -- ecn_ed448_mulgen_get_batch: generator multiplication fused with the affine export
// (csrc/ed28.h ed448_mulgen_get_one), the call sequence ecnXXXgen + ecnXXXmul + ecnXXXget that opens EdDSA key generation
// and signing in the reference (ed448.c:167-184, 196-199).  Fixed-base table: generated/comb_ED448.h.
#include "../../include/modarith_amd.h"
#include "capi_common.h"
#include "generated/params_X448.h"
#include "generated/comb_ED448.h"
#include "kernels.h"
#include "ed28.h"
#include "ed28l_k.h"

namespace ma {

// COMB_ED448_WINDOWS windows of COMB_ED448_W bits x 2^(W-1) multiples x coordinates x limbs, the same for every lane: constant address space, wave-uniform indices
__constant__ int32_t comb_ed448[] = { COMB_ED448_VALUES };
struct CombED448 {
    static constexpr int W = COMB_ED448_W, NW = COMB_ED448_WINDOWS;
    static __device__ __forceinline__ int32_t get(int idx) { return comb_ed448[idx]; }
};

// e*G through the fixed-base table, ONE scalar per lane.  SELF = false (the product path, round 5): the Edwards (X : Y : Z) go to the shared
// inversion of csrc/edlad_k.h (one inversion per up to 32 records; rounds 2-4 shared one between the TWO scalars of a lane, the first
// result parked in LDS, and spilled 14 registers in that epilogue).  SELF = true: the inversion in the kernel, for callers without
// scratch (a stream under capture): no workspace, no second kernel.
template <bool SELF>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_ed448_mulgen(const unsigned char* e, unsigned char* xb, unsigned char* yb, int* sign, size_t first, size_t n, Ed28lWs ws) {
    using P = P_X448;
    auto tt = [&]() {
        unsigned l = threadIdx.x;
        asm volatile("" : "+v"(l));
        return (size_t)blockIdx.x * 64 + l;
    };
    if (tt() >= n) return;
    Ed28::Ext R;
    {
        spint ew[7];
        load_be_record<P>(e, first + tt(), ew);
        ed448_mulgen_acc<CombED448>(ew, R);
    }
    if constexpr (SELF) {
        using F = Fe28;
        uint32_t zi[16], ax[16];
        spint w[7];
        F::invert(R.Z, zi);
        F::mul_k(R.X, zi, ax);
        F::to_words(ax, w);
        const int sx = (int)(w[0] & 1);
        if (xb) store_be_record<P>(xb, first + tt(), w);
        F::mul_k(R.Y, zi, ax);
        F::to_words(ax, w);
        if (yb) store_be_record<P>(yb, first + tt(), w);
        if (sign) sign[first + tt()] = !yb ? (int)(w[0] & 1) : (!xb ? sx : 0);
    } else {
        ws.store_xyz(tt(), R.X, R.Y, R.Z);
    }
}

// rfc7748() on the base point u = 5: [k](5) = Y^2 / X^2 of k*G on ED448 (ed28.h x448_base_one), little-endian 56-byte records as
// rfc7748_X448_batch takes them.  SELF = false: numerator and denominator to the shared inversion (a zero denominator -- the clamped
// key 4q, X = 0 -- enters the shared product as 1 and its result leaves as 0, the ladder's answer: fe_finish.h FeBatchDiv).
template <bool SELF>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_x448_base(const uint64_t* bk, uint64_t* bv, size_t first, size_t n, Ed28lWs ws) {
    auto tt = [&]() {
        unsigned l = threadIdx.x;
        asm volatile("" : "+v"(l));
        return (size_t)blockIdx.x * 64 + l;
    };
    if (tt() >= n) return;
    using F = Fe28;
    Ed28::Ext R;
    {
        uint64_t kw[7];
        static_for<0, 7>([&](auto K) { kw[K] = bk[(first + tt()) * 7 + K]; });
        kw[0] &= ~3ull;                                     // clamp (rfc7748.c:135-141)
        kw[6] |= 0x8000000000000000ull;
        ed448_mulgen_acc<CombED448>(kw, R);
    }
    uint32_t x2[16], y2[16];
    F::sqr_k(R.X, x2);
    F::sqr_k(R.Y, y2);
    if constexpr (SELF) {
        uint32_t xi[16], u[16];
        uint64_t ow[7];
        F::invert(x2, xi);                                  // (0^(p-2) = 0: X = 0 gives 0, as the ladder does)
        F::mul_k(y2, xi, u);
        F::to_words(u, ow);
        static_for<0, 7>([&](auto K) { bv[(first + tt()) * 7 + K] = ow[K]; });
    } else {
        ws.store_nd(tt(), y2, x2);
    }
}
// the quotient as a little-endian record of the caller's output array
struct SinkLE448 {
    uint64_t* bv;
    size_t first;
    MA_DEV void operator()(size_t e, uint64_t* bw, uint64_t*) const { static_for<0, 7>([&](auto K) { bv[(first + e) * 7 + K] = bw[K]; }); }
};

// round 5, the ladder form (csrc/ed28l.h): f*Q by the Montgomery ladder with the recovered Edwards point in extended coordinates, e*G
// added through the constant table; the inversions in front and behind are shared (csrc/edlad_k.h)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_ed448_lad_gen2(const unsigned char* e, const unsigned char* f, size_t first, Ed28lWs ws) {
    // the record index as a fresh value at each use (a 64-bit index carried across the 448 + 113 loop iterations is two registers the
    // kernel does not have: they were its last two spilled ones)
    auto tt = [&]() {
        unsigned l = threadIdx.x;
        asm volatile("" : "+v"(l));
        return (size_t)blockIdx.x * 64 + l;
    };
    if (tt() >= ws.m) return;
    using L = Ed28Lad;
    uint32_t x2[16], z2[16], x3[16], z3[16];
    bool f_odd;
    {
        spint fw[7];
        load_be_record<P_X448>(f, first + tt(), fw);
        f_odd = (fw[0] & 1) != 0;
        uint32_t u[16];
        ws.load_u(tt(), u);
        L::ladder(fw, u, x2, z2, x3, z3);
    }
    const uint32_t fl = ws.flags[tt()];
    Ed28::Ext R;
    L::recover([&](uint32_t* o) { ws.load_u(tt(), o); }, [&](uint32_t* o) { (void)ws.load_w(tt(), o); }, fl, f_odd, x2, z2, x3, z3, R, true);
    spint ew[7];
    load_be_record<P_X448>(e, first + tt(), ew);
    ed448_mulgen_acc<CombED448, false>(ew, R);           // += e*G through the fixed-base table
    ws.store_xyz(tt(), R.X, R.Y, R.Z);
}

}  // namespace ma

using namespace ma;

extern "C" size_t ecn_ed448_mulgen2_get_workspace_bytes(size_t n) { return ed28l_workspace_bytes(n); }

extern "C" int ecn_ed448_mulgen2_get_batch(const char* e, const char* f, const ma_spint* Q, char* x, char* y, int* sign, size_t n, size_t ld,
                                           void* workspace, size_t workspace_bytes, void* st) {
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(f) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 7u) {
        set_error("ecn mulgen2_get: byte records must be 8-byte aligned");
        return (int)hipErrorInvalidValue;
    }
    hipStream_t s = (hipStream_t)st;
    EdLadScratch ws(workspace, workspace_bytes, ed28l_workspace_bytes(n), 8, s);
    if (!ws.p) {
        set_error(std::string("ecn mulgen2_get: no usable workspace -- " + std::string(ws.why) + " (pass ecn_ed448_mulgen2_get_workspace_bytes(n) bytes; the library's own scratch pool is not available while the stream is being captured)"));
        return (int)hipErrorInvalidValue;
    }
    const unsigned char *eb = reinterpret_cast<const unsigned char*>(e), *fb = reinterpret_cast<const unsigned char*>(f);
    edlad_pipeline<LadT448, 2>(Q, ld, reinterpret_cast<unsigned char*>(x), reinterpret_cast<unsigned char*>(y), sign, n, ws.p, s,
                               [&](size_t first, size_t m, const Ed28lWs& w) { k_ed448_lad_gen2<<<(unsigned)((m + 63) / 64), 64, 0, s>>>(eb, fb, first, w); });
    return check_launch("ecn mulgen2_get (ladder form)");
}

extern "C" int ecn_ed448_mulgen_get_batch(const char* e, char* x, char* y, int* sign, size_t n, void* st) {
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 7u) {
        set_error("ecn mulgen_get: byte records must be 8-byte aligned");
        return (int)hipErrorInvalidValue;
    }
    hipStream_t s = (hipStream_t)st;
    const unsigned char* eb = reinterpret_cast<const unsigned char*>(e);
    unsigned char *xb = reinterpret_cast<unsigned char*>(x), *yb = reinterpret_cast<unsigned char*>(y);
    EdLadScratch ws(nullptr, 0, ed28l_workspace_bytes(n), 8, s);          // this entry point has no workspace argument: the library's scratch pool
    for (size_t first = 0; first < n; first += EDLAD_CHUNK) {
        const size_t m = n - first < EDLAD_CHUNK ? n - first : EDLAD_CHUNK;
        if (ws.p) {
            Ed28lWs w(ws.p, m);
            k_ed448_mulgen<false><<<(unsigned)((m + 63) / 64), 64, 0, s>>>(eb, xb, yb, sign, first, m, w);
            edlad_export<LadT448, 3>(w, xb, yb, sign, first, s);
        } else {
            k_ed448_mulgen<true><<<(unsigned)((m + 63) / 64), 64, 0, s>>>(eb, xb, yb, sign, first, m, Ed28lWs(nullptr, m));
        }
    }
    return check_launch("ecn mulgen_get");
}

// bv = [bk](5): rfc7748(bk, base, bv) for a batch of private keys, on the fixed-base table of ED448 (4-isogenous to curve448)
extern "C" int rfc7748_X448_base_batch(const char* bk, char* bv, size_t n, void* st) {
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(bk) | reinterpret_cast<uintptr_t>(bv)) & 7u) {
        set_error("rfc7748 base: byte records must be 8-byte aligned");
        return (int)hipErrorInvalidValue;
    }
    hipStream_t s = (hipStream_t)st;
    const uint64_t* kb = reinterpret_cast<const uint64_t*>(bk);
    uint64_t* vb = reinterpret_cast<uint64_t*>(bv);
    EdLadScratch ws(nullptr, 0, ed28l_workspace_bytes(n), 8, s);
    for (size_t first = 0; first < n; first += EDLAD_CHUNK) {
        const size_t m = n - first < EDLAD_CHUNK ? n - first : EDLAD_CHUNK;
        if (ws.p) {
            Ed28lWs w(ws.p, m);
            k_x448_base<false><<<(unsigned)((m + 63) / 64), 64, 0, s>>>(kb, vb, first, m, w);
            size_t L;
            int rounds;
            edlad_rounds(m, &L, &rounds);
            k_fe_batch_div<Fe28, 16, 7, SinkLE448, 4><<<(unsigned)((L + 63) / 64), 64, 0, s>>>(w.A, w.B, w.B, w.wc, m, L, rounds, SinkLE448{vb, first});
        } else {
            k_x448_base<true><<<(unsigned)((m + 63) / 64), 64, 0, s>>>(kb, vb, first, m, Ed28lWs(nullptr, m));
        }
    }
    return check_launch("rfc7748 base");
}
