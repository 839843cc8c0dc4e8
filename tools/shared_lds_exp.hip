// tools/shared_lds_exp.hip -- EXPERIMENT (GPU box): where should the shared multiplicand of modmuls live?  The north star suggests
// LDS staging; the product kernel (csrc/kernels.h k_mul_shared) takes it by value, i.e. in SGPRs.  Three variants over the same
// 2^24-element X25519 batch, split products, two elements per lane:
//   sgpr   the product kernel                      (b0 in the kernel arguments; its halves are scalar registers)
//   lds    b0 staged through LDS by each workgroup (global -> LDS once per workgroup, ds_read broadcast per lane)
//   gptr   b0 behind a device pointer              (uniform address: scalar loads, then as sgpr)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-codegenprepare-mul24=0 tools/shared_lds_exp.hip -o tools/shared_lds_exp.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../modarith_amd/csrc/kernels.h"
#include "../modarith_amd/csrc/generated/params_X25519.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
namespace ma {
using P = P_X25519;
__global__ __launch_bounds__(BLOCK) void k_lds(const spint* a, const spint* b0g, spint* c, size_t nthreads, size_t ld) {
    __shared__ spint sb[P::N];
    if (threadIdx.x < P::N) sb[threadIdx.x] = b0g[threadIdx.x];
    __syncthreads();
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < nthreads; t += (size_t)gridDim.x * BLOCK) {
        spint x[2][P::N], z[2][P::N], b[P::N];
        load_soa<P, 2>(a, ld, t, x);
        static_for<0, P::N>([&](auto I) { b[I] = sb[I]; });            // per-lane LDS reads of the common operand
        Field<P, true>::modmul(x[0], b, z[0]);
        Field<P, true>::modmul(x[1], b, z[1]);
        store_soa<P, 2>(c, ld, t, z);
    }
}
__global__ __launch_bounds__(BLOCK) void k_gptr(const spint* a, const spint* __restrict__ b0g, spint* c, size_t nthreads, size_t ld) {
    spint b[P::N];
    static_for<0, P::N>([&](auto I) { b[I] = b0g[I]; });                // uniform address: scalar loads
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < nthreads; t += (size_t)gridDim.x * BLOCK) {
        spint x[2][P::N], z[2][P::N];
        load_soa<P, 2>(a, ld, t, x);
        Field<P, true>::modmul(x[0], b, z[0]);
        Field<P, true>::modmul(x[1], b, z[1]);
        store_soa<P, 2>(c, ld, t, z);
    }
}
}
using namespace ma;
int main() {
    const size_t n = (size_t)1 << 24, nt = n / 2;
    spint *a, *c, *c2, *b0g;
    CK(hipMalloc(&a, n * 40)); CK(hipMalloc(&c, n * 40)); CK(hipMalloc(&c2, n * 40)); CK(hipMalloc(&b0g, 40));
    spint* h = (spint*)malloc(n * 40);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n * 5; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = s & ((1ull << 51) - 1); }
    CK(hipMemcpy(a, h, n * 40, hipMemcpyHostToDevice));
    Elem<P> b0; for (int i = 0; i < 5; i++) b0.l[i] = h[i * 7 + 3];
    CK(hipMemcpy(b0g, b0.l, 40, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto bench = [&](const char* name, auto launch, spint* out) {
        double best = 1e9, sum = 0;
        for (int i = 0; i < 3; i++) launch(out);
        for (int rep = 0; rep < 7; rep++) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; i++) launch(out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10; best = ms < best ? ms : best; sum += ms;
        }
        printf("%-6s  %.4f ms best, %.4f ms mean   %.0f GB/s (80 B per element)\n", name, best, sum / 7, 80.0 * n / (best * 1e-3) / 1e9);
    };
    for (int round = 0; round < 2; round++) {
        bench("sgpr", [&](spint* o) { k_mul_shared<P, 2, true><<<4096, BLOCK>>>(a, b0, o, nt, n, n); }, c);
        bench("lds", [&](spint* o) { k_lds<<<4096, BLOCK>>>(a, b0g, o, nt, n); }, c2);
        bench("gptr", [&](spint* o) { k_gptr<<<4096, BLOCK>>>(a, b0g, o, nt, n); }, c2);
    }
    spint* h2 = (spint*)malloc(n * 40);
    CK(hipMemcpy(h, c, n * 40, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2, c2, n * 40, hipMemcpyDeviceToHost));
    printf("outputs equal: %s\n", memcmp(h, h2, n * 40) == 0 ? "yes" : "NO");
    return 0;
}
