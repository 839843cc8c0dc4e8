// tools/ecn_exp.hip -- A/B harness for the scalar-multiplication kernels of the curve layer (csrc/curve.h): builds the
// kernels of ONE curve from a chosen source directory, times ecn mul / mul2 / mul2_exact on a batch of random scalars and
// prints a digest of the projective limbs, so that variants can be compared for speed AND for identical results in one
// GPU call.  Not part of the library; the library's own tests pin the limbs to the reference (tests/test_gpu_curveref.py).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-codegenprepare-mul24=0 -I<csrc dir> \
//         -DCURVE_HDR='"generated/curve_ED25519.h"' -DCURVE_CLASS='ma::Edwards<ma::C_ED25519>' -DNEWWS=1 \
//         tools/ecn_exp.hip -o tools/ecn_exp_<tag>.bin
//   ./ecn_exp_<tag>.bin [log2 n] [ops: 1 = mul, 2 = mul2, 4 = mul2x, 8 = gen]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include CURVE_HDR
#include "edwards.h"
#include "weierstrass.h"
#ifdef USE_FH51
#include "fh51.h"
#endif
#ifdef USE_FH56
#include "fh56.h"
#endif

using namespace ma;
using E = CURVE_CLASS;
#ifdef CURVE_MUL_CLASS
using EM = CURVE_MUL_CLASS;       // the class the scalar-multiplication kernels are built from (resident form of the field)
#else
using EM = E;
#endif

#ifndef EXP_GUARD
#define EXP_GUARD 0          // 1: the fast class with the limb-budget vote (csrc/curve.h), as the library launches it
#endif
#ifndef EXP_REPS
#define EXP_REPS 2
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_fill(unsigned char* e, size_t bytes, unsigned long long seed) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < bytes / 8; t += (size_t)gridDim.x * blockDim.x)
        reinterpret_cast<unsigned long long*>(e)[t] = splitmix64_at(seed, t);
}
__global__ void k_digest(const spint* p, size_t words, unsigned long long* out) {
    unsigned long long acc = 0;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < words; t += (size_t)gridDim.x * blockDim.x)
        acc += splitmix64_at(p[t], t);
    atomicAdd(out, acc);
}

// one wave on a second stream: waits `delay` wall-clock ticks, then reads the shader-clock counter (s_memtime) against the constant-rate
// wall clock (s_memrealtime) over `window` ticks -- the clock the part holds WHILE the kernel beside it runs (as modarith_amd/clock.py)
__global__ __launch_bounds__(64) void k_clk(unsigned long long* out, unsigned long long delay, unsigned long long window) {
    const unsigned long long w0 = wall_clock64();
    while (wall_clock64() - w0 < delay) __builtin_amdgcn_s_sleep(32);
    const unsigned long long wa = wall_clock64(), ca = clock64();
    while (wall_clock64() - wa < window) __builtin_amdgcn_s_sleep(32);
    const unsigned long long wb = wall_clock64(), cb = clock64();
    if (threadIdx.x == 0) { out[0] = cb - ca; out[1] = wb - wa; }
}
struct ClockProbe {
    hipStream_t s2;
    unsigned long long* d;
    int khz = 0;
    ClockProbe() {
        CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        CK(hipMalloc(&d, 16));
        CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0));
    }
    // call right after launching the kernel to observe (whose duration was `ms` on the run before)
    void start(float ms) {
        const unsigned long long ticks = (unsigned long long)(ms * 1e-3 * khz * 1e3);
        k_clk<<<1, 64, 0, s2>>>(d, ticks / 4, ticks / 2);
    }
    double ghz() {
        unsigned long long h[2];
        CK(hipStreamSynchronize(s2));
        CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        return h[1] ? (double)h[0] / (double)h[1] * khz / 1e6 : 0.0;
    }
};

__global__ void k_nop(spint* p) { if (p == nullptr) p[0] = 0; }
template <class Crv>
__global__ __launch_bounds__(64) void k_vote_only(spint* Pb, size_t n, size_t ld) {
    for (size_t base = (size_t)blockIdx.x * 64; base < n; base += (size_t)gridDim.x * 64) {
        if (base + threadIdx.x >= n) continue;
        if (__all(Crv::limbs_ok(Pb, ld, base + threadIdx.x))) continue;
        Pb[base + threadIdx.x] = 0;
    }
}
static unsigned long long digest(const spint* p, size_t words) {
    unsigned long long *d, h = 0;
    CK(hipMalloc(&d, 8));
    CK(hipMemset(d, 0, 8));
    k_digest<<<1024, 256>>>(p, words, d);
    CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    CK(hipFree(d));
    return h;
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 18;
    const int ops = argc > 2 ? atoi(argv[2]) : 1;
    const size_t n = (size_t)1 << lg;
    constexpr int N = E::N, NB = E::NB;
    const size_t rec = (size_t)E::NW * 8;        // scalar records padded to whole words (NB % 8 != 0: byte records)
    unsigned char *e, *f;
    spint *P, *Q, *R, *ws;
    CK(hipMalloc(&e, n * rec)); CK(hipMalloc(&f, n * rec));
    CK(hipMalloc(&P, n * 3 * N * 8)); CK(hipMalloc(&Q, n * 3 * N * 8)); CK(hipMalloc(&R, n * 3 * N * 8));
    k_fill<<<1024, 256>>>(e, n * rec, 1);
    k_fill<<<1024, 256>>>(f, n * rec, 2);
    size_t lanes = (n + 63) / 64 * 64;
    const size_t cap = (size_t)(4 * MA_MUL_WPS) * 256 * 64;
    if (lanes > cap) lanes = cap;
    CK(hipMalloc(&ws, lanes * E::TABLE_WORDS * 8));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    float ms;
    const char* name = CURVE_NAME;
    k_ed_op<E, ED_GEN><<<(unsigned)((n + 255) / 256), 256>>>(nullptr, P, n, n);
    k_ed_op<E, ED_GEN><<<(unsigned)((n + 255) / 256), 256>>>(nullptr, Q, n, n);
    CK(hipDeviceSynchronize());
    if (ops & 8) {
        CK(hipEventRecord(t0));
        k_ed_op<E, ED_GEN><<<(unsigned)((n + 255) / 256), 256>>>(nullptr, Q, n, n);
        CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1)); CK(hipEventElapsedTime(&ms, t0, t1));
        printf("%s gen   n=2^%d  %.3f ms  %.3e per s  digest %016llx\n", name, lg, ms, n / (ms * 1e-3), digest(Q, n * 3 * N));
    }
    // spread the points: P_j = f_j * G (untimed warm-up of the mul kernel as well)
    k_ed_mul<EM, EXP_GUARD><<<(unsigned)(lanes / 64), 64>>>(f, Q, n, n, ws);
    CK(hipDeviceSynchronize());
    if (ops & 1) {
        for (int r = 0; r < EXP_REPS; r++) {
            CK(hipMemcpy(P, Q, n * 3 * N * 8, hipMemcpyDeviceToDevice));
            CK(hipEventRecord(t0));
            k_ed_mul<EM, EXP_GUARD><<<(unsigned)(lanes / 64), 64>>>(e, P, n, n, ws);
#if EXP_GUARD == 1 && defined(EXP_EXACT_BEHIND)
#if EXP_EXACT_BEHIND == 1
            k_ed_mul<typename exact_class<EM>::type, -1><<<(unsigned)(lanes / 64), 64>>>(e, P, n, n, ws);     // as the library: the exact class behind the fast one
#elif EXP_EXACT_BEHIND == 2
            k_ed_mul<typename exact_class<EM>::type, -1><<<256, 64>>>(e, P, n, n, ws);                         // a small grid
#elif EXP_EXACT_BEHIND == 3
            k_nop<<<(unsigned)(lanes / 64), 64>>>(P);                                                          // an empty kernel of the same grid
#elif EXP_EXACT_BEHIND == 4
            k_vote_only<EM><<<(unsigned)(lanes / 64), 64>>>(P, n, n);                                          // the vote alone, fast class, no LDS, few registers
#endif
#endif
            CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1)); CK(hipEventElapsedTime(&ms, t0, t1));
            printf("%s mul   n=2^%d  %.3f ms  %.3e per s  digest %016llx\n", name, lg, ms, n / (ms * 1e-3), digest(P, n * 3 * N));
        }
        {   // once more with the shader-clock probe beside it
            ClockProbe cp;
            CK(hipMemcpy(P, Q, n * 3 * N * 8, hipMemcpyDeviceToDevice));
            CK(hipDeviceSynchronize());
            k_ed_mul<EM, EXP_GUARD><<<(unsigned)(lanes / 64), 64>>>(e, P, n, n, ws);
            cp.start(ms);
            CK(hipDeviceSynchronize());
            printf("%s mul   shader clock under the kernel %.3f GHz\n", name, cp.ghz());
        }
    }
    if (ops & 6) {
        k_ed_op<E, ED_GEN><<<(unsigned)((n + 255) / 256), 256>>>(nullptr, P, n, n);
        CK(hipDeviceSynchronize());
    }
    if (ops & 2) {
        for (int r = 0; r < 2; r++) {
            CK(hipEventRecord(t0));
            k_ed_mul2<EM, EXP_GUARD><<<(unsigned)(lanes / 64), 64>>>(e, P, f, Q, R, n, n, ws);
            CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1)); CK(hipEventElapsedTime(&ms, t0, t1));
            printf("%s mul2  n=2^%d  %.3f ms  %.3e per s  digest %016llx\n", name, lg, ms, n / (ms * 1e-3), digest(R, n * 3 * N));
        }
        {
            ClockProbe cp;
            k_ed_mul2<EM, EXP_GUARD><<<(unsigned)(lanes / 64), 64>>>(e, P, f, Q, R, n, n, ws);
            cp.start(ms);
            CK(hipDeviceSynchronize());
            printf("%s mul2  shader clock under the kernel %.3f GHz\n", name, cp.ghz());
        }
    }
    if (ops & 4) {
        for (int r = 0; r < 2; r++) {
            CK(hipEventRecord(t0));
            k_ed_mul2x<EM><<<(unsigned)(lanes / 64), 64>>>(e, P, f, Q, R, n, n, ws);
            CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1)); CK(hipEventElapsedTime(&ms, t0, t1));
            printf("%s mul2x n=2^%d  %.3f ms  %.3e per s  digest %016llx\n", name, lg, ms, n / (ms * 1e-3), digest(R, n * 3 * N));
        }
    }
    (void)NB;
    return 0;
}
