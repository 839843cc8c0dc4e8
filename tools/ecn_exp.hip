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
    k_ed_mul<EM><<<(unsigned)(lanes / 64), 64>>>(f, Q, n, n, ws);
    CK(hipDeviceSynchronize());
    if (ops & 1) {
        for (int r = 0; r < 2; r++) {
            CK(hipMemcpy(P, Q, n * 3 * N * 8, hipMemcpyDeviceToDevice));
            CK(hipEventRecord(t0));
            k_ed_mul<EM><<<(unsigned)(lanes / 64), 64>>>(e, P, n, n, ws);
            CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1)); CK(hipEventElapsedTime(&ms, t0, t1));
            printf("%s mul   n=2^%d  %.3f ms  %.3e per s  digest %016llx\n", name, lg, ms, n / (ms * 1e-3), digest(P, n * 3 * N));
        }
    }
    if (ops & 6) {
        k_ed_op<E, ED_GEN><<<(unsigned)((n + 255) / 256), 256>>>(nullptr, P, n, n);
        CK(hipDeviceSynchronize());
    }
    if (ops & 2) {
        for (int r = 0; r < 2; r++) {
            CK(hipEventRecord(t0));
            k_ed_mul2<EM><<<(unsigned)(lanes / 64), 64>>>(e, P, f, Q, R, n, n, ws);
            CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1)); CK(hipEventElapsedTime(&ms, t0, t1));
            printf("%s mul2  n=2^%d  %.3f ms  %.3e per s  digest %016llx\n", name, lg, ms, n / (ms * 1e-3), digest(R, n * 3 * N));
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
