// modarith_amd/csrc/capi_common.hip -- prime-independent part of the C-ABI shim (include/modarith_amd.h):
// device/memory utilities, error reporting, AoS<->SoA converters, field_info.
#include "../../include/modarith_amd.h"
#include "capi_common.h"
#include "kernels.h"
#include <atomic>
#include <string.h>

namespace ma {

static thread_local std::string g_err;
static thread_local const char* g_last_launch = "";

void set_error(const std::string& msg) { g_err = msg; }

int check_launch(const char* what) {
    g_last_launch = what;                      // string literals only: modarith_amd_last_launch()
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(std::string(what) + ": " + hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

static int env_int(const char* name, int dflt, int lo, int hi) {
    const char* s = getenv(name);
    if (!s || !*s) return dflt;
    long v = strtol(s, nullptr, 10);
    if (v < lo) v = lo;
    if (v > hi) v = hi;
    return (int)v;
}

int max_blocks() {
    static int v = env_int("MA_MAX_BLOCKS", 256 * 16, 1, 1 << 24);
    return v;
}
int max_blocks_tiled() {
    static int v = env_int("MA_MAX_BLOCKS_TILED", 1 << 16, 1, 1 << 24);
    return v;
}
int ladder_block() {
    static int v = env_int("MA_LADDER_BLOCK", 64, 64, 256) / 64 * 64;    // the ladder kernels carry __launch_bounds__(256)
    return v;
}

bool force_fast() {
    static bool v = [] { const char* s = getenv("MA_FORCE_FAST"); return s && *s == '1'; }();
    return v;
}
bool force_exact() {
    static bool v = [] { const char* s = getenv("MA_FORCE_EXACT"); return s && *s == '1'; }();
    return v;
}
bool inv_simul() {
    static bool v = [] { const char* s = getenv("MA_INV_SIMUL"); return !(s && *s == '0'); }();
    return v;
}
bool ladder_split() {
    static bool v = [] { const char* s = getenv("MA_LADDER_SPLIT"); return !(s && *s == '0'); }();
    return v;
}
bool ladder_use_field() {
    static bool v = [] { const char* s = getenv("MA_LADDER_IMPL"); return s && strcmp(s, "field") == 0; }();
    return v;
}

static std::atomic<int> g_sticky{0};
static thread_local int t_sticky = 0;
int thread_status() { return t_sticky; }
void clear_thread_status() { t_sticky = 0; }
void fail(const char* what, hipError_t e) {
    set_error(std::string(what) + " failed: " + hipGetErrorString(e));
    int zero = 0;
    const int code = e == hipSuccess ? (int)hipErrorUnknown : (int)e;
    g_sticky.compare_exchange_strong(zero, code);
    if (t_sticky == 0) t_sticky = code;
    (void)hipGetLastError();
}
int sticky_status() { return g_sticky.load(); }
void clear_sticky_status() { g_sticky.store(0); }

unsigned char* Staging::get(int d) {
    if (!dev[d]) {
        // MA_TEST_STAGING_FAIL=1 (tests only): behave as if the device had no memory left for the staging buffer
        static const bool test_fail = [] { const char* s = getenv("MA_TEST_STAGING_FAIL"); return s && *s == '1'; }();
        hipError_t e = test_fail ? hipErrorOutOfMemory : hipMalloc((void**)&dev[d], BYTES);
        if (e != hipSuccess) {
            dev[d] = nullptr;
            fail("hipMalloc(staging)", e);
            return nullptr;
        }
    }
    return dev[d];
}
StageBase::StageBase() {
    int d = 0;
    hipError_t e = hipGetDevice(&d);
    if (e != hipSuccess) { fail("hipGetDevice(staging)", e); bad = true; return; }
    if (d < 0 || d >= Staging::MAX_DEVICES) { fail("staging: device index out of range", hipErrorInvalidDevice); bad = true; return; }
    lock = std::unique_lock<std::mutex>(staging().mu[d]);
    base = staging().get(d);
    bad = base == nullptr;
}
void* StageBase::take(size_t bytes) {
    if (bad) return nullptr;
    void* d = base + used;
    used += (bytes + 15) / 16 * 16;
    if (used > Staging::BYTES) { fail("staging overflow", hipErrorOutOfMemory); bad = true; return nullptr; }
    return d;
}
void StageBase::h2d(void* d, const void* h, size_t b) {
    if (nsrc < 8) src[nsrc++] = Src{static_cast<const unsigned char*>(h), b};
    if (bad) return;
    hipError_t e = hipMemcpy(d, h, b, hipMemcpyHostToDevice);
    if (e != hipSuccess) { fail("hipMemcpy(h2d)", e); bad = true; }
}
void StageBase::d2h(void* h, const void* d, size_t b) {
    if (!bad) {
        hipError_t e = hipMemcpy(h, d, b, hipMemcpyDeviceToHost);
        if (e == hipSuccess) return;
        fail("hipMemcpy(d2h)", e);
        bad = true;
    }
    const unsigned char* hp = static_cast<const unsigned char*>(h);
    for (int i = 0; i < nsrc; i++)
        if (hp < src[i].p + src[i].b && src[i].p < hp + b) return;         // also an input of this call: untouched
    memset(h, 0, b);
}
void StageBase::check(int rc, const char* what) {
    if (rc != 0) { fail(what, (hipError_t)rc); bad = true; }
}
// stream-ordered scratch of the library's own (the batched ladders' split form): one memory pool per device, created on
// first use, that keeps what it has handed out once (release threshold = max), so that a resident caller pays for the
// workspace of its largest batch once and not per call.  nullptr when pools are unavailable: callers then take the
// self-contained kernels.
static std::mutex g_pool_mu;
static hipMemPool_t g_pools[Staging::MAX_DEVICES] = {};
static bool g_pool_failed[Staging::MAX_DEVICES] = {};
// what the pool keeps cached between calls: enough for the split ladder of a 2^24-record batch (1.2 GiB) -- beyond that, freed
// blocks go back to the driver at the next synchronisation; modarith_amd_scratch_trim() releases on demand
static constexpr uint64_t POOL_KEEP_BYTES = (uint64_t)1280 << 20;
void* scratch_alloc(size_t bytes, hipStream_t s) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= Staging::MAX_DEVICES) return nullptr;
    // never allocate while a capture is in progress that this call could invalidate: on the stream itself, or -- for the legacy
    // default stream (s == nullptr), which a global-mode capture of another thread's stream also covers -- anywhere in the process
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (cs != hipStreamCaptureStatusNone) return nullptr;
    hipMemPool_t pool = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        if (!g_pools[d] && !g_pool_failed[d]) {
            hipMemPoolProps props = {};
            props.allocType = hipMemAllocationTypePinned;
            props.handleTypes = hipMemHandleTypeNone;
            props.location.type = hipMemLocationTypeDevice;
            props.location.id = d;
            if (hipMemPoolCreate(&g_pools[d], &props) == hipSuccess) {
                uint64_t keep = POOL_KEEP_BYTES;
                (void)hipMemPoolSetAttribute(g_pools[d], hipMemPoolAttrReleaseThreshold, &keep);
            } else {
                g_pools[d] = nullptr;
                g_pool_failed[d] = true;
                (void)hipGetLastError();
            }
        }
        pool = g_pools[d];
    }
    if (!pool) return nullptr;
    void* p = nullptr;
    if (hipMallocFromPoolAsync(&p, bytes, pool, s) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
// give cached scratch of the current device back to the driver, keeping at most keep_bytes
int scratch_trim(size_t keep_bytes) {
    int d = 0;
    hipError_t e = hipGetDevice(&d);
    if (e != hipSuccess) return (int)e;
    if (d < 0 || d >= Staging::MAX_DEVICES) return (int)hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_pool_mu);
    if (!g_pools[d]) return 0;
    e = hipMemPoolTrimTo(g_pools[d], keep_bytes);
    if (e != hipSuccess) { set_error(std::string("hipMemPoolTrimTo: ") + hipGetErrorString(e)); return (int)e; }
    return 0;
}
void scratch_free(void* p, hipStream_t s) {
    if (p) (void)hipFreeAsync(p, s);
}
Staging& staging() {
    static Staging s;
    return s;
}
// element-major (AoS, spint x[n][N] as CPU callers hold elements) <-> limb-interleaved SoA (flat or tiled: kernels.h Ld); any limb count
static __device__ __forceinline__ size_t soa_off(Ld L, int nlimbs, size_t j) {
    return (((j >> L.s) * (size_t)nlimbs) << L.s) + (j & ((((size_t)1) << L.s) - 1));
}
static __global__ __launch_bounds__(BLOCK) void k_aos2soa(const spint* aos, spint* soa, size_t n, int nlimbs, Ld L) {
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (size_t)gridDim.x * BLOCK)
        for (int i = 0; i < nlimbs; i++) soa[soa_off(L, nlimbs, t) + (size_t)i * L.ld] = aos[t * (size_t)nlimbs + i];
}
static __global__ __launch_bounds__(BLOCK) void k_soa2aos(const spint* soa, spint* aos, size_t n, int nlimbs, Ld L) {
    for (size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (size_t)gridDim.x * BLOCK)
        for (int i = 0; i < nlimbs; i++) aos[t * (size_t)nlimbs + i] = soa[soa_off(L, nlimbs, t) + (size_t)i * L.ld];
}
// The same conversions through LDS, for 16-byte-aligned buffers, even strides and up to 14 limbs: a workgroup of 256 lanes moves
// a chunk of 512 elements.  Both sides of the transposition then run as whole 16-byte-per-lane coalesced accesses -- the SoA rows
// two elements per lane, the chunk's AoS image (512 * nlimbs consecutive words) as one linear stretch -- instead of one side
// striding nlimbs words from lane to lane (soa->aos 2.8 -> TB/s class; profiles/history/r03_elementwise_and_converters.log).  LDS image: sh[e * S + i] with
// S = nlimbs | 1 words per element (an odd pitch keeps the strided side at two-way bank conflicts).
constexpr int CONV_CHUNK = 512;
template <bool TO_SOA>
static __global__ __launch_bounds__(BLOCK) void k_convert_lds(const spint* src, spint* dst, size_t n, int nlimbs, Ld L) {
    extern __shared__ spint sh[];
    const int S = nlimbs | 1, t = threadIdx.x;
    for (size_t c0 = (size_t)blockIdx.x * CONV_CHUNK; c0 < n; c0 += (size_t)gridDim.x * CONV_CHUNK) {
        const size_t cnt = (n - c0 < (size_t)CONV_CHUNK) ? n - c0 : (size_t)CONV_CHUNK;
        const size_t total = cnt * (size_t)nlimbs;                      // words of the chunk's AoS image
        const spint* aos_in = src + c0 * (size_t)nlimbs;                // (TO_SOA)
        spint* aos_out = dst + c0 * (size_t)nlimbs;                     // (!TO_SOA)
        // ---- the SoA side: lane t holds elements 2t, 2t+1 of the chunk
        const size_t j = c0 + 2 * (size_t)t;
        const bool two = 2 * (size_t)t + 1 < cnt, one = 2 * (size_t)t < cnt;
        const size_t so = soa_off(L, nlimbs, j);
        if constexpr (!TO_SOA) {
            if (two) {
                for (int i = 0; i < nlimbs; i++) {
                    const spint2 v = __builtin_nontemporal_load(reinterpret_cast<const spint2*>(src + so + (size_t)i * L.ld));
                    sh[(2 * t) * S + i] = v.x;
                    sh[(2 * t + 1) * S + i] = v.y;
                }
            } else if (one) {
                for (int i = 0; i < nlimbs; i++) sh[(2 * t) * S + i] = src[so + (size_t)i * L.ld];
            }
        } else {
            // ---- the AoS side, linear: lane t takes words 2t, 2t+1, then + 512, ...; (e, i) = divmod(w, nlimbs) kept incrementally
            int e = (2 * t) / nlimbs, i = (2 * t) % nlimbs;
            const int de = CONV_CHUNK / nlimbs, di = CONV_CHUNK % nlimbs;
            for (size_t w = 2 * (size_t)t; w < total; w += CONV_CHUNK) {
                int e1 = e, i1 = i + 1;
                if (i1 == nlimbs) { i1 = 0; e1++; }
                if (w + 1 < total) {
                    const spint2 v = __builtin_nontemporal_load(reinterpret_cast<const spint2*>(aos_in + w));
                    sh[e * S + i] = v.x;
                    sh[e1 * S + i1] = v.y;
                } else {
                    sh[e * S + i] = aos_in[w];
                }
                e += de; i += di;
                if (i >= nlimbs) { i -= nlimbs; e++; }
            }
        }
        __syncthreads();
        if constexpr (!TO_SOA) {
            int e = (2 * t) / nlimbs, i = (2 * t) % nlimbs;
            const int de = CONV_CHUNK / nlimbs, di = CONV_CHUNK % nlimbs;
            for (size_t w = 2 * (size_t)t; w < total; w += CONV_CHUNK) {
                int e1 = e, i1 = i + 1;
                if (i1 == nlimbs) { i1 = 0; e1++; }
                if (w + 1 < total) {
                    spint2 v;
                    v.x = sh[e * S + i];
                    v.y = sh[e1 * S + i1];
                    __builtin_nontemporal_store(v, reinterpret_cast<spint2*>(aos_out + w));
                } else {
                    aos_out[w] = sh[e * S + i];
                }
                e += de; i += di;
                if (i >= nlimbs) { i -= nlimbs; e++; }
            }
        } else {
            if (two) {
                for (int i = 0; i < nlimbs; i++) {
                    spint2 v;
                    v.x = sh[(2 * t) * S + i];
                    v.y = sh[(2 * t + 1) * S + i];
                    __builtin_nontemporal_store(v, reinterpret_cast<spint2*>(dst + so + (size_t)i * L.ld));
                }
            } else if (one) {
                for (int i = 0; i < nlimbs; i++) dst[so + (size_t)i * L.ld] = sh[(2 * t) * S + i];
            }
        }
        __syncthreads();
    }
}
static bool conv_fast_ok(const void* aos, const void* soa, int nlimbs, Ld L) {
    return nlimbs <= 14 && L.ld % 2 == 0 && aligned16(aos) && aligned16(soa);
}
static unsigned conv_grid(size_t n, bool tiled) {
    size_t b = (n + CONV_CHUNK - 1) / CONV_CHUNK;
    const size_t cap = (size_t)(tiled ? max_blocks_tiled() : 4 * max_blocks());
    return (unsigned)(b < cap ? b : cap);
}

// ld >= n: flat; ld < n: tiles of ld elements (a power of two >= 128)
static bool conv_ld(size_t n, size_t ld, Ld* L) {
    if (ld >= n) { *L = Ld(ld); return true; }
    if (ld < 128 || (ld & (ld - 1)) != 0) return false;
    *L = Ld(ld, (unsigned)__builtin_ctzll((unsigned long long)ld));
    return true;
}

static int wrap(hipError_t e, const char* what) {
    if (e != hipSuccess) {
        set_error(std::string(what) + ": " + hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

}  // namespace ma

using namespace ma;

// one wave: wait `delay` wall-clock ticks, then the shader-clock counter against the wall clock over `window` ticks
static __global__ __launch_bounds__(64) void k_sclk_probe(uint64_t* out, uint64_t delay, uint64_t window) {
    const uint64_t w0 = wall_clock64();
    while (wall_clock64() - w0 < delay) __builtin_amdgcn_s_sleep(32);
    const uint64_t wa = wall_clock64(), ca = clock64();
    while (wall_clock64() - wa < window) __builtin_amdgcn_s_sleep(32);
    const uint64_t wb = wall_clock64(), cb = clock64();
    if (threadIdx.x == 0) { out[0] = cb - ca; out[1] = wb - wa; }
}

extern "C" {

int modarith_amd_abi_version(void) { return MODARITH_AMD_ABI; }
const char* modarith_amd_last_error(void) { return g_err.c_str(); }
const char* modarith_amd_last_launch(void) { return g_last_launch; }
int modarith_amd_status(void) { return sticky_status(); }
void modarith_amd_clear_status(void) { clear_sticky_status(); }
int modarith_amd_thread_status(void) { return thread_status(); }
void modarith_amd_clear_thread_status(void) { clear_thread_status(); }
int modarith_amd_wall_clock_khz(void) {
    int d = 0, khz = 0;
    if (hipGetDevice(&d) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, d) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return khz;
}
int modarith_amd_sclk_probe(uint64_t* out, unsigned delay_us, unsigned window_us, void* st) {
    const int khz = modarith_amd_wall_clock_khz();
    if (khz <= 0) { set_error("sclk_probe: no wall-clock rate for this device"); return (int)hipErrorNotSupported; }
    k_sclk_probe<<<1, 64, 0, (hipStream_t)st>>>(out, (uint64_t)delay_us * (uint64_t)khz / 1000, (uint64_t)window_us * (uint64_t)khz / 1000);
    return check_launch("sclk_probe");
}
int modarith_amd_scratch_trim(size_t keep_bytes) { return scratch_trim(keep_bytes); }
// the limb stride a caller without a layout of its own should use (include/modarith_amd.h, TILED): tiles of 4096 elements once
// the batch holds two of them, flat rows below
size_t modarith_amd_recommended_ld(size_t n) { return n >= 2 * (size_t)4096 ? (size_t)4096 : n; }
// the same with the shape of the field: 8-limb (and longer) elements stream best on tiles of 8192 -- at 2^24 elements the median of 24
// operand placements is 0.808 / 0.823 of the HBM peak (modsqr / modmul of X448) against 0.801 / 0.821 on 4096, and the slow placements are
// one in 24 instead of three or four; 5-limb fields keep 4096 (0.801 / 0.809 against 0.792 / 0.804); 16384 is worse for both
// (profiles/r06_tile_shape_sweep.log).  No tile size moves the slow placements themselves (0.74-0.78): they belong to the memory system.
size_t modarith_amd_recommended_ld_for(size_t n, int nlimbs) {
    const size_t t = nlimbs >= 8 ? (size_t)8192 : (size_t)4096;
    return n >= 2 * t ? t : modarith_amd_recommended_ld(n);
}
size_t modarith_amd_batch_words(size_t n, int nlimbs, size_t ld) {
    if (nlimbs < 1 || ld == 0) return 0;
    if (ld >= n) return (size_t)nlimbs * ld;                       // flat
    return (n + ld - 1) / ld * ld * (size_t)nlimbs;               // whole tiles
}
int modarith_amd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int modarith_amd_set_device(int dev) { return wrap(hipSetDevice(dev), "hipSetDevice"); }
int modarith_amd_malloc(void** dptr, size_t bytes) { return wrap(hipMalloc(dptr, bytes), "hipMalloc"); }
int modarith_amd_free(void* dptr) { return wrap(hipFree(dptr), "hipFree"); }
int modarith_amd_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
    return wrap(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream), "hipMemcpyAsync(h2d)");
}
int modarith_amd_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
    return wrap(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream), "hipMemcpyAsync(d2h)");
}
int modarith_amd_sync(void* stream) { return wrap(hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize"); }
int modarith_amd_stream_create(void** stream) {
    hipStream_t s = nullptr;
    int rc = wrap(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate");
    *stream = (void*)s;
    return rc;
}
int modarith_amd_stream_destroy(void* stream) { return wrap(hipStreamDestroy((hipStream_t)stream), "hipStreamDestroy"); }
int modarith_amd_stream_wait(void* stream, void* other) {
    hipEvent_t ev;
    int rc = wrap(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
    if (rc) return rc;
    rc = wrap(hipEventRecord(ev, (hipStream_t)other), "hipEventRecord");
    if (!rc) rc = wrap(hipStreamWaitEvent((hipStream_t)stream, ev, 0), "hipStreamWaitEvent");
    hipError_t e = hipEventDestroy(ev);      // safe: destruction is deferred until the event has completed
    return rc ? rc : wrap(e, "hipEventDestroy");
}
int modarith_amd_host_alloc(void** hptr, size_t bytes) { return wrap(hipHostMalloc(hptr, bytes, hipHostMallocDefault), "hipHostMalloc"); }
int modarith_amd_host_free(void* hptr) { return wrap(hipHostFree(hptr), "hipHostFree"); }

int modarith_amd_aos_to_soa(const ma_spint* aos, ma_spint* soa, size_t n, int nlimbs, size_t ld, void* stream) {
    if (n == 0) return 0;
    Ld L;
    if (nlimbs < 1 || nlimbs > 64 || !conv_ld(n, ld, &L)) { set_error("aos_to_soa: need 1 <= nlimbs <= 64 and ld >= n (flat) or ld a power of two >= 128 (tiles)"); return (int)hipErrorInvalidValue; }
    if (conv_fast_ok(aos, soa, nlimbs, L))
        k_convert_lds<true><<<conv_grid(n, L.s != 63), BLOCK, (size_t)CONV_CHUNK * (nlimbs | 1) * sizeof(spint), (hipStream_t)stream>>>(aos, soa, n, nlimbs, L);
    else
        k_aos2soa<<<grid_for(n), BLOCK, 0, (hipStream_t)stream>>>(aos, soa, n, nlimbs, L);
    return check_launch("aos_to_soa");
}
int modarith_amd_soa_to_aos(const ma_spint* soa, ma_spint* aos, size_t n, int nlimbs, size_t ld, void* stream) {
    if (n == 0) return 0;
    Ld L;
    if (nlimbs < 1 || nlimbs > 64 || !conv_ld(n, ld, &L)) { set_error("soa_to_aos: need 1 <= nlimbs <= 64 and ld >= n (flat) or ld a power of two >= 128 (tiles)"); return (int)hipErrorInvalidValue; }
    if (conv_fast_ok(aos, soa, nlimbs, L))
        k_convert_lds<false><<<conv_grid(n, L.s != 63), BLOCK, (size_t)CONV_CHUNK * (nlimbs | 1) * sizeof(spint), (hipStream_t)stream>>>(soa, aos, n, nlimbs, L);
    else
        k_soa2aos<<<grid_for(n), BLOCK, 0, (hipStream_t)stream>>>(soa, aos, n, nlimbs, L);
    return check_launch("soa_to_aos");
}

int modarith_amd_field_info(const char* prime, int* nlimbs, int* radix, int* nbits, int* nbytes, int* montgomery) {
    struct Row { const char* name; int nl, rx, nb, by, mo; };
    static const Row rows[] = {
#include "generated/field_table.inc"
    };
    for (const Row& r : rows) {
        if (strcmp(prime, r.name) == 0) {
            if (nlimbs) *nlimbs = r.nl;
            if (radix) *radix = r.rx;
            if (nbits) *nbits = r.nb;
            if (nbytes) *nbytes = r.by;
            if (montgomery) *montgomery = r.mo;
            return 1;
        }
    }
    return 0;
}

}  // extern "C"
