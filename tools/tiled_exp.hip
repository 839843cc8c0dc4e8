// tools/tiled_exp.hip -- EXPERIMENT (GPU box): does a tiled limb-interleaved layout [tile][limb][lane] remove the dependence of the
// modmul streaming rate on where the driver placed the three 640 MiB arrays?  K operand triples are allocated one after the
// other; each is timed through the flat layout buf[limb * n + j] (the product kernel) and through tiles of 2^T elements
// buf[((j >> T) * N + limb) << T | (j & (2^T - 1))] over the SAME memory (all limbs random 51-bit: in contract either way).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-codegenprepare-mul24=0 tools/tiled_exp.hip -o tools/tiled_exp.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../modarith_amd/csrc/kernels.h"
#include "../modarith_amd/csrc/generated/params_X25519.h"
#include "../modarith_amd/csrc/generated/params_X448.h"
#ifndef EXP_P
#define EXP_P P_X25519
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
namespace ma {
using P = EXP_P;
#ifndef EXP_STORE
#define EXP_STORE 0
#endif
// store flavours for the experiment: 0 = the product's non-temporal store; 1 = sc1; 2 = sc0 sc1; 3 = nt sc1; 4 = plain
__device__ __forceinline__ void st_exp(spint2* p, spint2 v) {
#if EXP_STORE == 0
    st_stream(p, v);
#elif EXP_STORE == 1
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#elif EXP_STORE == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
#elif EXP_STORE == 3
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
#else
    *p = v;
#endif
}
#ifndef EXP_LOAD
#define EXP_LOAD 0
#endif
__device__ __forceinline__ spint2 ld_exp(const spint2* p) {
#if EXP_LOAD == 0
    return ld_stream(p);
#elif EXP_LOAD == 1
    spint2 v; asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v;
#else
    return *p;
#endif
}
// SWZ 0: workgroup b takes chunk b (consecutive chunks go round-robin over the 8 XCDs); SWZ 1: XCD x (= b % 8) takes the
// contiguous eighth [x * G/8, (x+1) * G/8) of the chunks of each grid-stride pass
template <int T, int SWZ = 0>
__global__ __launch_bounds__(BLOCK) void k_tiled(const spint* a, const spint* b, spint* c, size_t nthreads) {
    const size_t blk = SWZ ? (size_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : blockIdx.x;
    for (size_t t = blk * BLOCK + threadIdx.x; t < nthreads; t += (size_t)gridDim.x * BLOCK) {
        const size_t j = 2 * t, tile = j >> T, off = j & (((size_t)1 << T) - 1);
        const size_t base = ((tile * P::N) << T) + off;
        spint x[2][P::N], y[2][P::N], z[2][P::N];
        static_for<0, P::N>([&](auto I) {
            spint2 v = ld_stream(reinterpret_cast<const spint2*>(a + base + ((size_t)I << T)));
            x[0][I] = v.x; x[1][I] = v.y;
        });
        static_for<0, P::N>([&](auto I) {
            spint2 v = ld_stream(reinterpret_cast<const spint2*>(b + base + ((size_t)I << T)));
            y[0][I] = v.x; y[1][I] = v.y;
        });
        OpMulAuto<P>::apply(x[0], y[0], z[0]);
        OpMulAuto<P>::apply(x[1], y[1], z[1]);
        static_for<0, P::N>([&](auto I) {
            spint2 v; v.x = z[0][I]; v.y = z[1][I];
            st_stream(reinterpret_cast<spint2*>(c + base + ((size_t)I << T)), v);
        });
    }
}
// one 2*BS-element chunk per workgroup of BS lanes (no grid-stride loop)
template <int T, int BS>
__global__ __launch_bounds__(BS) void k_tiled_bs(const spint* a, const spint* b, spint* c, size_t nthreads) {
    const size_t t = (size_t)blockIdx.x * BS + threadIdx.x;
    if (t >= nthreads) return;
    const size_t j = 2 * t, tile = j >> T, off = j & (((size_t)1 << T) - 1);
    const size_t base = ((tile * P::N) << T) + off;
    spint x[2][P::N], y[2][P::N], z[2][P::N];
    static_for<0, P::N>([&](auto I) {
        spint2 v = ld_stream(reinterpret_cast<const spint2*>(a + base + ((size_t)I << T)));
        x[0][I] = v.x; x[1][I] = v.y;
    });
    static_for<0, P::N>([&](auto I) {
        spint2 v = ld_stream(reinterpret_cast<const spint2*>(b + base + ((size_t)I << T)));
        y[0][I] = v.x; y[1][I] = v.y;
    });
    OpMulAuto<P>::apply(x[0], y[0], z[0]);
    OpMulAuto<P>::apply(x[1], y[1], z[1]);
    static_for<0, P::N>([&](auto I) {
        spint2 v; v.x = z[0][I]; v.y = z[1][I];
        st_exp(reinterpret_cast<spint2*>(c + base + ((size_t)I << T)), v);
    });
}
template <int T, bool SQR>
__global__ __launch_bounds__(BLOCK) void k_tiled_un(const spint* a, spint* c, size_t nthreads) {
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < nthreads; t += (size_t)gridDim.x * BLOCK) {
        const size_t j = 2 * t, tile = j >> T, off = j & (((size_t)1 << T) - 1);
        const size_t base = ((tile * P::N) << T) + off;
        spint x[2][P::N], z[2][P::N];
        static_for<0, P::N>([&](auto I) {
            spint2 v = ld_stream(reinterpret_cast<const spint2*>(a + base + ((size_t)I << T)));
            x[0][I] = v.x; x[1][I] = v.y;
        });
        if constexpr (SQR) { OpSqrAuto<P>::apply(x[0], z[0]); OpSqrAuto<P>::apply(x[1], z[1]); }
        else { static_for<0, P::N>([&](auto I) { z[0][I] = x[0][I]; z[1][I] = x[1][I]; }); }
        static_for<0, P::N>([&](auto I) {
            spint2 v; v.x = z[0][I]; v.y = z[1][I];
            st_stream(reinterpret_cast<spint2*>(c + base + ((size_t)I << T)), v);
        });
    }
}
__global__ void k_fill(spint* p, size_t n, uint64_t seed) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = seed + (t + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        p[t] = z & ((1ull << P::RADIX) - 1);
    }
}
}
using namespace ma;
int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 6;
    const size_t n = (size_t)1 << 24, nt = n / 2;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto tm = [&](auto launch) {
        for (int i = 0; i < 3; i++) launch();
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; i++) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return 24.0 * P::N * n / (ms / 10 * 1e-3) / 1e9;
    };
    printf("%s  triple   mul: grid-stride g32768 | one chunk per workgroup: BS=64  128  256  512  1024 | T=13 BS=256  T=11 BS=256  (GB/s)\n", P::NAME);
    for (int k = 0; k < K; k++) {
        spint *a, *b, *c;
        CK(hipMalloc(&a, n * 8 * P::N)); CK(hipMalloc(&b, n * 8 * P::N)); CK(hipMalloc(&c, n * 8 * P::N));
        k_fill<<<4096, 256>>>(a, n * P::N, 1 + k); k_fill<<<4096, 256>>>(b, n * P::N, 100 + k);
        CK(hipDeviceSynchronize());
        double r[8];
        r[0] = tm([&] { k_tiled<12><<<32768, BLOCK>>>(a, b, c, nt); });
        r[1] = tm([&] { k_tiled_bs<12, 64><<<(unsigned)(nt / 64), 64>>>(a, b, c, nt); });
        r[2] = tm([&] { k_tiled_bs<12, 128><<<(unsigned)(nt / 128), 128>>>(a, b, c, nt); });
        r[3] = tm([&] { k_tiled_bs<12, 256><<<(unsigned)(nt / 256), 256>>>(a, b, c, nt); });
        r[4] = tm([&] { k_tiled_bs<12, 512><<<(unsigned)(nt / 512), 512>>>(a, b, c, nt); });
        r[5] = tm([&] { k_tiled_bs<12, 1024><<<(unsigned)(nt / 1024), 1024>>>(a, b, c, nt); });
        r[6] = tm([&] { k_tiled_bs<13, 256><<<(unsigned)(nt / 256), 256>>>(a, b, c, nt); });
        r[7] = tm([&] { k_tiled_bs<11, 256><<<(unsigned)(nt / 256), 256>>>(a, b, c, nt); });
        printf("%4d   %6.0f | %6.0f %6.0f %6.0f %6.0f %6.0f | %6.0f %6.0f\n", k, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
        // keep the memory (do not free): the next triple lands elsewhere
    }
    return 0;
}
