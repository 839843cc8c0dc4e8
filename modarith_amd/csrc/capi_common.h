// modarith_amd/csrc/capi_common.h -- shared host-side helpers of the C-ABI shim (launch geometry,
// error capture, scalar staging).  Host code only; no torch types anywhere in the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>

namespace ma {

void set_error(const std::string& msg);
int check_launch(const char* what);
int max_blocks();          // grid cap for grid-stride streaming kernels on flat batches (env MA_MAX_BLOCKS, 4096)
int max_blocks_tiled();    // the same for tiled batches (env MA_MAX_BLOCKS_TILED, 65536)
int ladder_block();        // workgroup size of the ladder kernel (env MA_LADDER_BLOCK)
bool force_fast();         // env MA_FORCE_FAST=1: element-wise modmul/modsqr/nres/redc/modinv on the FAST product path (tests)
bool force_exact();        // env MA_FORCE_EXACT=1: element-wise modmul/modsqr on the exact 128-bit products only (tests)
bool inv_simul();          // env MA_INV_SIMUL=0: modinv_<P>_batch never shares an inversion between elements
bool ladder_split();       // env MA_LADDER_SPLIT=0: the batched ladders never take the split form (one inversion per lane)
int scratch_trim(size_t keep_bytes);   // give the current device's cached scratch beyond keep_bytes back to the driver
bool ladder_use_field();   // env MA_LADDER_IMPL=field: X25519 ladder on the 5x51 field.c-form arithmetic

// tiled batches (kernels.h Ld) take a much larger cap: on tiles the streaming rate keeps rising with the number of
// workgroups up to one 512-element chunk per workgroup (X25519 modmul 5.96 / 6.16 / 6.39 / 6.58 TB/s at 4096 / 8192 / 16384 /
// 32768 workgroups, profiles/history/r03_tiled_exp_4_grid.log), whereas flat batches are insensitive to it
inline unsigned grid_for(size_t nthreads, int block = 256, bool tiled = false) {
    size_t b = (nthreads + (size_t)block - 1) / (size_t)block;
    size_t cap = (size_t)(tiled ? max_blocks_tiled() : max_blocks());
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Failures of the host side that the reference's void / predicate signatures cannot return (the scalar _ct entry points: no staging
// buffer, a failed copy, a failed launch): recorded, never fatal.  fail() sets the calling thread's modarith_amd_last_error() text and
// the process-wide sticky status modarith_amd_status() (first error code since the last modarith_amd_clear_status()); the entry point
// then returns without launching, its outputs zero-filled.  (Up to round 4 these paths called abort() inside the shared library.)
void fail(const char* what, hipError_t e);
int sticky_status();
void clear_sticky_status();
int thread_status();            // the same per calling thread: what THIS thread's scalar calls have recorded
void clear_thread_status();

// staging area for the scalar (_ct) entry points: one small device buffer PER DEVICE, each guarded by its own mutex (scalar calls on
// different devices do not serialise each other)
struct Staging {
    static constexpr int MAX_DEVICES = 64;
    std::mutex mu[MAX_DEVICES];
    unsigned char* dev[MAX_DEVICES] = {};         // one buffer per device of this process, made on first use
    static constexpr size_t BYTES = 512 * 1024;   // holds the scalar ecn mul2 window tables (64 lanes x up to 486 words)
    unsigned char* get(int d);                    // nullptr (and fail()) when the buffer cannot be made; call with mu[d] held
};
Staging& staging();
// One scalar call's use of the staging buffer: locks the current device's buffer for its lifetime.  bad = no buffer (no device, failed
// allocation, overflow) or a failed step: every later step is skipped, d2h() zero-fills what the caller expects back.
struct StageBase {
    std::unique_lock<std::mutex> lock;
    unsigned char* base = nullptr;
    size_t used = 0;
    bool bad = false;
    // host ranges this call has uploaded FROM: when the call fails, a result that would land on one of them (an in-place operand,
    // modmul(z2, E, z2)) is left as it is -- only pure outputs are zero-filled (round-5 advisor: a failed modnsqr / ecn dbl zeroed its input)
    struct Src { const unsigned char* p; size_t b; } src[8];
    int nsrc = 0;
    // a predicate's answer: -1, not the reference's 0 / 1, when the call failed
    int answer(int r) const { return bad ? -1 : r; }
    StageBase();
    void* take(size_t bytes);
    void h2d(void* d, const void* h, size_t b);
    void d2h(void* h, const void* d, size_t b);
    void check(int rc, const char* what);
};
void* scratch_alloc(size_t bytes, hipStream_t s);   // stream-ordered scratch from the library's own pool; nullptr if unavailable
void scratch_free(void* p, hipStream_t s);

// the caller's workspace when it holds `need` bytes behind its next `align`-aligned address (the pointer is rounded up HERE: a workspace
// function whose kernels want more than 8-byte alignment reports need + align - 1, and any pointer will do), else stream-ordered scratch of
// the library's own pool (released in stream order when this object goes); p = nullptr when neither is to be had (a stream under capture
// and no usable caller workspace): `why` then says what was wrong with the caller's
struct EdLadScratch {
    void* p = nullptr;
    void* own = nullptr;
    hipStream_t s;
    const char* why = "no caller workspace";
    EdLadScratch(void* workspace, size_t workspace_bytes, size_t need, size_t align, hipStream_t s_) : s(s_) {
        if (workspace) {
            const uintptr_t a = reinterpret_cast<uintptr_t>(workspace), up = (a + align - 1) & ~(uintptr_t)(align - 1);
            if (workspace_bytes >= need + (size_t)(up - a)) { p = reinterpret_cast<void*>(up); return; }
            why = workspace_bytes >= need ? "the caller's workspace is too small once its address is rounded up to the required alignment"
                                          : "the caller's workspace is too small";
        }
        p = own = scratch_alloc(need, s);
    }
    ~EdLadScratch() { if (own) scratch_free(own, s); }
};

}  // namespace ma
