// tools/ladder_occ_bench.hip -- EXPERIMENT (GPU box): the X25519 ladder kernel at 3 waves per SIMD (134 VGPRs, as the compiler
// allocates under __launch_bounds__(256)) against 4 (amdgpu_waves_per_eu(4,4): 114 VGPRs) and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-codegenprepare-mul24=0 tools/ladder_occ_bench.hip -o tools/ladder_occ_bench.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define MA_LADDER_FE26 1
#include "../modarith_amd/csrc/fe26.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
namespace ma {
#define BODY                                                                                                               \
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {          \
        uint64_t kw[4], uw[4], xw[4], zw[4];                                                                               \
        static_for<0, 4>([&](auto K) { kw[K] = bk[t * 4 + K]; });                                                          \
        static_for<0, 4>([&](auto K) { uw[K] = bu[t * 4 + K]; });                                                          \
        uint32_t x2[10], z2[10];                                                                                           \
        x25519_fe26_ladder(kw, uw, x2, z2);                                                                                \
        Fe26::to_words(x2, xw);                                                                                            \
        Fe26::to_words(z2, zw);                                                                                            \
        static_for<0, 4>([&](auto K) { bv[t * 4 + K] = xw[K]; });                                                          \
        static_for<0, 4>([&](auto K) { wz[(size_t)K * n + t] = zw[K]; });                                                  \
    }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_w4(const uint64_t* bk, const uint64_t* bu, uint64_t* bv, uint64_t* wz, size_t n) { BODY }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_w2(const uint64_t* bk, const uint64_t* bu, uint64_t* bv, uint64_t* wz, size_t n) { BODY }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_w5(const uint64_t* bk, const uint64_t* bu, uint64_t* bv, uint64_t* wz, size_t n) { BODY }
}
using namespace ma;
template <class K> void run(const char* name, K kern, uint64_t* bk, uint64_t* bu, uint64_t* bv, uint64_t* wz, size_t n) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        kern<<<4096, 64>>>(bk, bu, bv, wz, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s %8.3f ms  %.4e scalars/s\n", name, ms, n / (ms * 1e-3));
    }
}
int main() {
    const size_t n = 1 << 22;
    uint64_t *bk, *bu, *bv, *wz;
    CK(hipMalloc(&bk, n * 32)); CK(hipMalloc(&bu, n * 32)); CK(hipMalloc(&bv, n * 32)); CK(hipMalloc(&wz, n * 32));
    uint64_t* h = (uint64_t*)malloc(n * 32);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n * 4; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = s; }
    CK(hipMemcpy(bk, h, n * 32, hipMemcpyHostToDevice));
    for (size_t i = 0; i < n * 4; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = s; }
    CK(hipMemcpy(bu, h, n * 32, hipMemcpyHostToDevice));
    run("3 waves/SIMD (product)", k_x25519_fe26_xz, bk, bu, bv, wz, n);
    uint64_t* ref = (uint64_t*)malloc(n * 32); CK(hipMemcpy(ref, bv, n * 32, hipMemcpyDeviceToHost));
    run("4 waves/SIMD", k_w4, bk, bu, bv, wz, n);
    CK(hipMemcpy(h, bv, n * 32, hipMemcpyDeviceToHost)); printf("  equal to product: %s\n", memcmp(h, ref, n * 32) == 0 ? "yes" : "NO");
    run("5 waves/SIMD", k_w5, bk, bu, bv, wz, n);
    CK(hipMemcpy(h, bv, n * 32, hipMemcpyDeviceToHost)); printf("  equal to product: %s\n", memcmp(h, ref, n * 32) == 0 ? "yes" : "NO");
    run("2 waves/SIMD", k_w2, bk, bu, bv, wz, n);
    return 0;
}
