// tools/fe29_bench.hip -- EXPERIMENT (GPU box): nine 29-bit limbs (tools/fe29_exp.h) against ten 25.5-bit limbs
// (csrc/fe26.h) for the X25519 ladder arithmetic: dependent chains of multiplications, of squarings and of whole ladder
// steps per lane, 8 waves per SIMD resident; prints ns per operation per lane-batch and writes the limbs of a few lanes
// (inputs and outputs) for the value check in tools/fe29_check.py.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-codegenprepare-mul24=0 tools/fe29_bench.hip -o tools/fe29_bench.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../modarith_amd/csrc/fe26.h"
#include "fe29_exp.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
using namespace ma;

__device__ __forceinline__ uint64_t sm64(uint64_t s, uint64_t t) {
    uint64_t z = s + (t + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// KIND 0: f = f*g  1: f = f^2  2: ladder step on (x2,z2,x3,z3) with x1 = g
template <class F, int NL, int BITS_E, int BITS_O, int KIND>
__global__ __launch_bounds__(256) void k_chain(uint32_t* io, int iters, size_t n) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    uint32_t f[NL], g[NL], x3[NL], z3[NL], z2[NL];
    for (int i = 0; i < NL; i++) {
        const int b = (i & 1) ? BITS_O : BITS_E;
        f[i] = (uint32_t)sm64(1, t * 64 + i) & ((1u << b) - 1);
        g[i] = (uint32_t)sm64(2, t * 64 + i) & ((1u << b) - 1);
        x3[i] = (uint32_t)sm64(3, t * 64 + i) & ((1u << b) - 1);
        z3[i] = (uint32_t)sm64(4, t * 64 + i) & ((1u << b) - 1);
        z2[i] = (uint32_t)sm64(5, t * 64 + i) & ((1u << b) - 1);
    }
    if (t < 64) for (int i = 0; i < NL; i++) { io[(t * 6 + 0) * 16 + i] = f[i]; io[(t * 6 + 1) * 16 + i] = g[i]; io[(t * 6 + 2) * 16 + i] = z2[i]; io[(t * 6 + 3) * 16 + i] = x3[i]; io[(t * 6 + 4) * 16 + i] = z3[i]; }
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        if constexpr (KIND == 0) F::mul(f, g, f);
        else if constexpr (KIND == 1) F::sqr(f, f);
        else {
            const bool sw = (sm64(6, t) >> (it & 63)) & 1;
            if constexpr (NL == 9) {
                F::step(sw, g, f, z2, x3, z3);
            } else {
                uint32_t g19[10];
                F::pre19(g, g19);
                uint32_t A[10], B[10], C[10], D[10], As[10], Bs[10], AA[10], BB[10], E[10];
                F::add(f, z2, A); F::add(x3, z3, C); F::sub(f, z2, B); F::sub(x3, z3, D);
                F::select(sw, A, C, As); F::select(sw, B, D, Bs);
                F::mul(D, A, D); F::mul(C, B, C); F::sqr(As, AA); F::sqr(Bs, BB);
                F::sub(D, C, z3); F::add(D, C, x3); F::sub(AA, BB, E);
                F::template mul_small_add<121665>(E, AA, z2);
                F::mul(z2, E, z2); F::sqr(x3, x3); F::sqr(z3, z3); F::mul(z3, g, g19, z3); F::mul(AA, BB, f);
            }
        }
    }
    uint32_t acc = 0;
    for (int i = 0; i < NL; i++) acc ^= f[i] ^ z2[i] ^ x3[i] ^ z3[i];
    if (t < 64) for (int i = 0; i < NL; i++) io[(t * 6 + 5) * 16 + i] = f[i];
    if (acc == 0x12345678u) io[0] = acc;      // keeps everything live
}

template <class F, int NL, int BE, int BO, int KIND>
double run(const char* name, int iters, uint32_t* io_d, uint32_t* io_h, FILE* fo) {
    const size_t n = (size_t)256 * 8 * 4 * 64;             // 8 waves per SIMD on every CU
    CK(hipMemset(io_d, 0, 64 * 6 * 16 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_chain<F, NL, BE, BO, KIND><<<n / 256, 256>>>(io_d, 16, n);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_chain<F, NL, BE, BO, KIND><<<n / 256, 256>>>(io_d, iters, n);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(io_h, io_d, 64 * 6 * 16 * 4, hipMemcpyDeviceToHost));
    const double ns_per_op_per_wave = ms * 1e6 / iters / 8.0;       // 8 waves share a SIMD: time one wave-op occupies the SIMD
    printf("%-22s %8.3f ms  %7.1f ns per wave-op per SIMD  (%.0f cycles at 2.4 GHz)   %.3e lane-ops/s\n", name, ms, ns_per_op_per_wave, ns_per_op_per_wave * 2.4,
           (double)n * iters / (ms * 1e-3));
    fprintf(fo, "%s %d %d %d\n", name, NL, KIND, iters);
    for (int t = 0; t < 64; t++) for (int r = 0; r < 6; r++) { for (int i = 0; i < NL; i++) fprintf(fo, "%u ", io_h[(t * 6 + r) * 16 + i]); fprintf(fo, "\n"); }
    return ms;
}

int main(int argc, char** argv) {
    const char* out = argc > 1 ? argv[1] : "fe29_check.txt";
    FILE* fo = fopen(out, "w");
    uint32_t* io_d; CK(hipMalloc(&io_d, 64 * 6 * 16 * 4));
    uint32_t* io_h = (uint32_t*)malloc(64 * 6 * 16 * 4);
    for (int rep = 0; rep < 2; rep++) {
        run<Fe26, 10, 26, 25, 0>("fe26 mul", 4096, io_d, io_h, fo);
        run<Fe29, 9, 29, 29, 0>("fe29 mul", 4096, io_d, io_h, fo);
        run<Fe26, 10, 26, 25, 1>("fe26 sqr", 4096, io_d, io_h, fo);
        run<Fe29, 9, 29, 29, 1>("fe29 sqr", 4096, io_d, io_h, fo);
        run<Fe26, 10, 26, 25, 2>("fe26 ladder step", 1024, io_d, io_h, fo);
        run<Fe29, 9, 29, 29, 2>("fe29 ladder step", 1024, io_d, io_h, fo);
    }
    fclose(fo);
    return 0;
}
