// tools/valubench.hip -- issue-rate micro-benchmark for the integer / fp64 VALU instructions a field
// multiplication can be built from on gfx950 (GPU box only).  Prints cycles per wave-instruction per
// SIMD at full occupancy (8 waves/SIMD) and with one wave per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/valubench.hip -o tools/valubench.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITER = 2048;   // loop trips
constexpr int UNR = 16;      // instructions per trip (8 independent chains x 2)

#define BODY8(stmt) stmt(0) stmt(1) stmt(2) stmt(3) stmt(4) stmt(5) stmt(6) stmt(7)

template <int KIND>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t seed, uint64_t* clk = nullptr) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + threadIdx.x * 7;
    // round 5: the shader-clock counter (s_memtime) and the constant-rate wall clock (s_memrealtime) around the loop, so that the
    // cost of an instruction comes out in CYCLES OF THE CLOCK THE PART HELD, not in nanoseconds x a nominal 2.4 GHz
    const uint64_t c0 = clock64(), w0 = wall_clock64();
    uint64_t acc[8];
    uint32_t r[8];
    double d[8];
    for (int i = 0; i < 8; i++) { acc[i] = a * (i + 1); r[i] = b + i; d[i] = 1.0 + i + threadIdx.x; }
    double da = 1.000001 + threadIdx.x, db = 0.999999;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if constexpr (KIND == 0) {
#define S(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(r[i]) : "vcc");
                BODY8(S)
#undef S
            } else if constexpr (KIND == 1) {
#define S(i) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
                BODY8(S)
#undef S
            } else if constexpr (KIND == 2) {
#define S(i) asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
                BODY8(S)
#undef S
            } else if constexpr (KIND == 3) {
#define S(i) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
                BODY8(S)
#undef S
            } else if constexpr (KIND == 4) {
#define S(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(da), "v"(db));
                BODY8(S)
#undef S
            } else if constexpr (KIND == 5) {
#define S(i) asm volatile("v_add_f64 %0, %1, %0" : "+v"(d[i]) : "v"(da));
                BODY8(S)
#undef S
            } else if constexpr (KIND == 6) {
#define S(i) asm volatile("v_add_co_u32 %0, vcc, %1, %0" : "+v"(r[i]) : "v"(a) : "vcc");
                BODY8(S)
#undef S
            } else if constexpr (KIND == 7) {
#define S(i) asm volatile("v_addc_co_u32 %0, vcc, %1, %0, vcc" : "+v"(r[i]) : "v"(a) : "vcc");
                BODY8(S)
#undef S
            } else if constexpr (KIND == 8) {
#define S(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) & 7]));
                BODY8(S)
#undef S
            } else if constexpr (KIND == 9) {
#define S(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
                BODY8(S)
#undef S
            } else if constexpr (KIND == 10) {
#define S(i) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(a));
                BODY8(S)
#undef S
            } else if constexpr (KIND == 11) {
#define S(i) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(r[i]) : "v"(a));
                BODY8(S)
#undef S
            } else if constexpr (KIND == 12) {
#define S(i) asm volatile("v_mul_hi_u32_u24 %0, %1, %0" : "+v"(r[i]) : "v"(a));
                BODY8(S)
#undef S
            } else if constexpr (KIND == 13) {
#define S(i) asm volatile("v_lshrrev_b64 %0, 13, %0" : "+v"(acc[i]));
                BODY8(S)
#undef S
            } else if constexpr (KIND == 14) {
#define S(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
                BODY8(S)
#undef S
            } else if constexpr (KIND == 15) {
#define S(i) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(d[i]) : "v"(da));
                BODY8(S)
#undef S
            } else if constexpr (KIND == 16) {
#define S(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(r[i]) : "vcc");
                BODY8(S)
#undef S
            } else if constexpr (KIND == 17) {
#define S(i) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(r[i]) : "v"(a) : );
                BODY8(S)
#undef S
            } else if constexpr (KIND == 18) {
#define S(i) asm volatile("v_alignbit_b32 %0, %1, %0, 7" : "+v"(r[i]) : "v"(a) : );
                BODY8(S)
#undef S
            } else if constexpr (KIND == 19) {
#define S(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
                BODY8(S)
#undef S
            }
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < 8; i++) s += acc[i] + r[i] + (uint64_t)d[i];
    const uint64_t c1 = clock64(), w1 = wall_clock64();
    if (clk && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND> void run(const char* name, uint64_t* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int configs[2][2] = {{256 * 8, 256}, {256, 256}};   // 8 waves/SIMD; 1 wave/SIMD
    double cyc[2], real[2], ghz[2];
    static uint64_t* clk = nullptr;
    static int khz = 0;
    if (!clk) { CK(hipMalloc(&clk, 16)); CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0)); }
    for (int c = 0; c < 2; c++) {
        int blocks = configs[c][0];
        k<KIND><<<blocks, 256>>>(out, 1); CK(hipDeviceSynchronize());
        for (int rep = 0; rep < 40; rep++) k<KIND><<<blocks, 256>>>(out, rep);      // (warm: let the clock settle under this instruction)
        CK(hipEventRecord(e0));
        for (int rep = 0; rep < 5; rep++) k<KIND><<<blocks, 256>>>(out, rep, clk);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        // waves per SIMD executed in sequence = blocks*4 waves / (256 CUs * 4 SIMDs)
        double waves_per_simd = blocks * 4.0 / 1024.0;
        double instr = (double)ITER * UNR * waves_per_simd;
        cyc[c] = ms * 1e-3 * 2.4e9 / instr;    // cycles (at nominal 2.4 GHz) per wave-instruction per SIMD
        uint64_t h[2];
        CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
        // one wave of the middle block: its lifetime in shader cycles / the instructions its SIMD issued meanwhile (all waves resident at once)
        real[c] = (double)h[0] / ((double)ITER * UNR * waves_per_simd);
        ghz[c] = h[1] ? (double)h[0] / (double)h[1] * khz * 1e3 / 1e9 : 0.0;
    }
    printf("%-22s  %6.2f cyc/instr @8 waves/SIMD   %6.2f cyc/instr @1 wave/SIMD   | at the measured clock: %5.2f (%.3f GHz)  %5.2f (%.3f GHz)\n",
           name, cyc[0], cyc[1], real[0], ghz[0], real[1], ghz[1]);
    fflush(stdout);
}

int main() {
    uint64_t* out; CK(hipMalloc(&out, 256 * 8 * 256 * 8));
    run<10>("v_mov_b32", out);
    run<9>("v_add_u32", out);
    run<14>("v_and_b32", out);
    run<6>("v_add_co_u32", out);
    run<7>("v_addc_co_u32", out);
    run<17>("v_cndmask_b32", out);
    run<18>("v_alignbit_b32", out);
    run<8>("v_lshl_add_u64", out);
    run<13>("v_lshrrev_b64", out);
    run<3>("v_mad_u32_u24", out);
    run<11>("v_mul_u32_u24", out);
    run<12>("v_mul_hi_u32_u24", out);
    run<1>("v_mul_lo_u32", out);
    run<2>("v_mul_hi_u32", out);
    run<0>("v_mad_u64_u32", out);
    run<16>("v_mad_i64_i32", out);
    run<19>("v_fma_f32", out);
    run<4>("v_fma_f64", out);
    run<5>("v_add_f64", out);
    run<15>("v_mul_f64", out);
    return 0;
}
