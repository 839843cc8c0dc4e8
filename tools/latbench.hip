// tools/latbench.hip -- dependent-chain latency of the instructions a field product is built from (GPU box): ONE serial chain per
// wave, at 1 / 2 / 3 / 4 / 8 waves per SIMD.  Complements tools/valubench.hip (8 independent chains per wave = issue rate).
//   hipcc --offload-arch=gfx950 -O3 tools/latbench.hip -o tools/latbench.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITER = 4096, UNR = 32;

// KIND 0: acc = a*b + acc (v_mad_u64_u32), serial   1: two interleaved serial chains   2: four chains   3: v_and_b32 serial
//      4: mad -> lshrrev_b64 -> mad (carry extraction pattern)   5: v_lshl_add_u64 serial   6: v_add_u32 serial
template <int KIND>
__global__ __launch_bounds__(64) void k(uint64_t* out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + threadIdx.x * 7, r = b;
    uint64_t acc0 = a, acc1 = b, acc2 = a + 1, acc3 = b + 1;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            if constexpr (KIND == 0) {
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b) : "vcc");
            } else if constexpr (KIND == 1) {
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b) : "vcc");
            } else if constexpr (KIND == 2) {
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc2) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc3) : "v"(a), "v"(b) : "vcc");
            } else if constexpr (KIND == 3) {
                asm volatile("v_and_b32 %0, %1, %0" : "+v"(r) : "v"(a));
            } else if constexpr (KIND == 4) {
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_lshrrev_b64 %0, 26, %0" : "+v"(acc0));
            } else if constexpr (KIND == 5) {
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc0) : "v"(acc1));
            } else {
                asm volatile("v_add_u32 %0, %1, %0" : "+v"(r) : "v"(a));
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc0 ^ acc1 ^ acc2 ^ acc3 ^ r;
}

template <int KIND>
void run(const char* name, int per_iter, uint64_t* out) {
    for (int waves : {1, 2, 3, 4, 8}) {
        const int blocks = 256 * 4 * waves;          // one 64-lane block per wave slot
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        k<KIND><<<blocks, 64>>>(out, 1); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        k<KIND><<<blocks, 64>>>(out, 2);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_wave = (double)ITER * UNR * per_iter;
        printf("%-34s %d waves/SIMD: %6.2f cycles per instruction per WAVE, %6.2f per SIMD (2.4 GHz)\n", name, waves,
               ms * 1e-3 * 2.4e9 / instr_per_wave, ms * 1e-3 * 2.4e9 / instr_per_wave / waves);
    }
}
int main() {
    uint64_t* out; CK(hipMalloc(&out, 256 * 4 * 8 * 64 * 8));
    run<0>("v_mad_u64_u32, 1 serial chain", 1, out);
    run<1>("v_mad_u64_u32, 2 chains", 2, out);
    run<2>("v_mad_u64_u32, 4 chains", 4, out);
    run<3>("v_and_b32, 1 serial chain", 1, out);
    run<6>("v_add_u32, 1 serial chain", 1, out);
    run<4>("mad -> lshrrev_b64 serial pair", 2, out);
    run<5>("v_lshl_add_u64, 1 serial chain", 1, out);
    return 0;
}
