// tools/membench.hip -- access-pattern micro-benchmark for the 3-array, N-plane streaming pattern of
// the element-wise field kernels (GPU box only).  Prints GB/s for layout / width / cache-policy variants.
//   hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench.bin && tools/membench.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
constexpr int N = 5;

template <int W, bool NT> __device__ __forceinline__ void ld(const u64* p, u64* v) {
    if constexpr (W == 1) { v[0] = NT ? __builtin_nontemporal_load(p) : *p; }
    else if constexpr (W == 2) {
        u64x2 t = NT ? __builtin_nontemporal_load((const u64x2*)p) : *(const u64x2*)p; v[0] = t.x; v[1] = t.y;
    } else {
        ld<2, NT>(p, v); ld<2, NT>(p + 2, v + 2);
    }
}
template <int W, bool NT> __device__ __forceinline__ void st(u64* p, const u64* v) {
    if constexpr (W == 1) { if (NT) __builtin_nontemporal_store(v[0], p); else *p = v[0]; }
    else if constexpr (W == 2) {
        u64x2 t; t.x = v[0]; t.y = v[1];
        if (NT) __builtin_nontemporal_store(t, (u64x2*)p); else *(u64x2*)p = t;
    } else { st<2, NT>(p, v); st<2, NT>(p + 2, v + 2); }
}

// plain SoA: plane stride ldp (elements); thread t handles elements [W*t, W*t+W)
template <int W, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_soa(const u64* a, const u64* b, u64* c, size_t nthreads, size_t ldp) {
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < nthreads; t += (size_t)gridDim.x * 256) {
        u64 x[N][W], y[N][W], z[N][W];
#pragma unroll
        for (int i = 0; i < N; i++) ld<W, NTL>(a + i * ldp + W * t, x[i]);
#pragma unroll
        for (int i = 0; i < N; i++) ld<W, NTL>(b + i * ldp + W * t, y[i]);
#pragma unroll
        for (int i = 0; i < N; i++)
#pragma unroll
            for (int w = 0; w < W; w++) z[i][w] = x[i][w] + y[(i + 1) % N][w];
#pragma unroll
        for (int i = 0; i < N; i++) st<W, NTS>(c + i * ldp + W * t, z[i]);
    }
}
// tiled (AoSoA): tile of T = 256*W elements: [tile][limb][lane]
template <int W, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_tiled(const u64* a, const u64* b, u64* c, size_t ntiles) {
    constexpr size_t T = 256 * W;
    for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const size_t base = tile * N * T + W * threadIdx.x;
        u64 x[N][W], y[N][W], z[N][W];
#pragma unroll
        for (int i = 0; i < N; i++) ld<W, NTL>(a + base + i * T, x[i]);
#pragma unroll
        for (int i = 0; i < N; i++) ld<W, NTL>(b + base + i * T, y[i]);
#pragma unroll
        for (int i = 0; i < N; i++)
#pragma unroll
            for (int w = 0; w < W; w++) z[i][w] = x[i][w] + y[(i + 1) % N][w];
#pragma unroll
        for (int i = 0; i < N; i++) st<W, NTS>(c + base + i * T, z[i]);
    }
}
// flat copy reference: c = a + b over 5n contiguous words (no plane structure)
template <int W>
__global__ __launch_bounds__(256) void k_flat(const u64* a, const u64* b, u64* c, size_t nthreads) {
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < nthreads; t += (size_t)gridDim.x * 256) {
        u64 x[W], y[W], z[W];
        ld<W, false>(a + W * t, x); ld<W, false>(b + W * t, y);
#pragma unroll
        for (int w = 0; w < W; w++) z[w] = x[w] + y[w];
        st<W, false>(c + W * t, z);
    }
}

template <class Fn> double timeit(Fn fn, int reps = 20) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) fn();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) fn();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    const size_t n = (size_t)1 << 24;
    const size_t maxpad = 1 << 16;
    const size_t words = N * (n + maxpad);
    u64 *a, *b, *c;
    CK(hipMalloc(&a, words * 8)); CK(hipMalloc(&b, words * 8)); CK(hipMalloc(&c, words * 8));
    CK(hipMemset(a, 1, words * 8)); CK(hipMemset(b, 2, words * 8)); CK(hipMemset(c, 0, words * 8));
    const double bytes = 3.0 * N * n * 8;
    auto rep = [&](const char* name, double ms) { printf("%-44s %8.4f ms  %8.1f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout); };
    int grids[] = {512, 1024, 2048, 4096, 16384};
    for (int g : grids) {
        char nm[128];
        snprintf(nm, sizeof nm, "flat W=2 grid=%d", g);
        rep(nm, timeit([&] { k_flat<2><<<g, 256>>>(a, b, c, N * n / 2); }));
    }
    rep("flat W=4 grid=2048", timeit([&] { k_flat<4><<<2048, 256>>>(a, b, c, N * n / 4); }));
    size_t pads[] = {0, 32, 64, 128, 256, 512, 1024, 2048 + 64, 4096 + 128, 8192 + 32 * 9, 65536 - 32};
    for (size_t pad : pads) {
        char nm[128];
        snprintf(nm, sizeof nm, "soa W=2 pad=%zu grid=2048", pad);
        rep(nm, timeit([&] { k_soa<2, false, false><<<2048, 256>>>(a, b, c, n / 2, n + pad); }));
    }
    for (int g : grids) {
        char nm[128];
        snprintf(nm, sizeof nm, "soa W=2 pad=0 grid=%d", g);
        rep(nm, timeit([&] { k_soa<2, false, false><<<g, 256>>>(a, b, c, n / 2, n); }));
        snprintf(nm, sizeof nm, "soa W=4 pad=0 grid=%d", g);
        rep(nm, timeit([&] { k_soa<4, false, false><<<g, 256>>>(a, b, c, n / 4, n); }));
        snprintf(nm, sizeof nm, "tiled W=2 grid=%d", g);
        rep(nm, timeit([&] { k_tiled<2, false, false><<<g, 256>>>(a, b, c, n / 512); }));
        snprintf(nm, sizeof nm, "tiled W=4 grid=%d", g);
        rep(nm, timeit([&] { k_tiled<4, false, false><<<g, 256>>>(a, b, c, n / 1024); }));
    }
    rep("soa W=2 nt-load", timeit([&] { k_soa<2, true, false><<<2048, 256>>>(a, b, c, n / 2, n); }));
    rep("soa W=2 nt-store", timeit([&] { k_soa<2, false, true><<<2048, 256>>>(a, b, c, n / 2, n); }));
    rep("soa W=2 nt-both", timeit([&] { k_soa<2, true, true><<<2048, 256>>>(a, b, c, n / 2, n); }));
    rep("soa W=4 nt-both", timeit([&] { k_soa<4, true, true><<<2048, 256>>>(a, b, c, n / 4, n); }));
    rep("tiled W=2 nt-both", timeit([&] { k_tiled<2, true, true><<<2048, 256>>>(a, b, c, n / 512); }));
    rep("tiled W=4 nt-both", timeit([&] { k_tiled<4, true, true><<<2048, 256>>>(a, b, c, n / 1024); }));
    rep("tiled W=4 nt-both g=16384", timeit([&] { k_tiled<4, true, true><<<16384, 256>>>(a, b, c, n / 1024); }));
    return 0;
}
