// modarith_amd/csrc/capi_common.hip -- prime-independent part of the C-ABI shim (include/modarith_amd.h):
// device/memory utilities, error reporting, AoS<->SoA converters, field_info.
#include "../../include/modarith_amd.h"
#include "capi_common.h"
#include "kernels.h"
#include <string.h>

namespace ma {

static thread_local std::string g_err;

void set_error(const std::string& msg) { g_err = msg; }

int check_launch(const char* what) {
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

unsigned char* Staging::get() {
    int d = 0;
    hipError_t e = hipGetDevice(&d);
    if (e != hipSuccess) die("hipGetDevice(staging)", e);
    if (d < 0 || d >= MAX_DEVICES) die("staging: device index out of range", hipErrorInvalidDevice);
    if (!dev[d]) {
        e = hipMalloc((void**)&dev[d], BYTES);
        if (e != hipSuccess) die("hipMalloc(staging)", e);
    }
    return dev[d];
}
// stream-ordered scratch of the library's own (the batched ladders' split form): one memory pool per device, created on
// first use, that keeps what it has handed out once (release threshold = max), so that a resident caller pays for the
// workspace of its largest batch once and not per call.  nullptr when pools are unavailable: callers then take the
// self-contained kernels.
void* scratch_alloc(size_t bytes, hipStream_t s) {
    static std::mutex mu;
    static hipMemPool_t pools[Staging::MAX_DEVICES] = {};
    static bool failed[Staging::MAX_DEVICES] = {};
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= Staging::MAX_DEVICES) return nullptr;
    hipMemPool_t pool = nullptr;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!pools[d] && !failed[d]) {
            hipMemPoolProps props = {};
            props.allocType = hipMemAllocationTypePinned;
            props.handleTypes = hipMemHandleTypeNone;
            props.location.type = hipMemLocationTypeDevice;
            props.location.id = d;
            if (hipMemPoolCreate(&pools[d], &props) == hipSuccess) {
                uint64_t keep = ~(uint64_t)0;
                (void)hipMemPoolSetAttribute(pools[d], hipMemPoolAttrReleaseThreshold, &keep);
            } else {
                pools[d] = nullptr;
                failed[d] = true;
                (void)hipGetLastError();
            }
        }
        pool = pools[d];
    }
    if (!pool) return nullptr;
    void* p = nullptr;
    if (hipMallocFromPoolAsync(&p, bytes, pool, s) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
void scratch_free(void* p, hipStream_t s) {
    if (p) (void)hipFreeAsync(p, s);
}
Staging& staging() {
    static Staging s;
    return s;
}
void die(const char* what, hipError_t e) {
    fprintf(stderr, "modarith_amd: %s failed: %s\n", what, hipGetErrorString(e));
    abort();
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

extern "C" {

int modarith_amd_abi_version(void) { return MODARITH_AMD_ABI; }
const char* modarith_amd_last_error(void) { return g_err.c_str(); }
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
    k_aos2soa<<<grid_for(n), BLOCK, 0, (hipStream_t)stream>>>(aos, soa, n, nlimbs, L);
    return check_launch("aos_to_soa");
}
int modarith_amd_soa_to_aos(const ma_spint* soa, ma_spint* aos, size_t n, int nlimbs, size_t ld, void* stream) {
    if (n == 0) return 0;
    Ld L;
    if (nlimbs < 1 || nlimbs > 64 || !conv_ld(n, ld, &L)) { set_error("soa_to_aos: need 1 <= nlimbs <= 64 and ld >= n (flat) or ld a power of two >= 128 (tiles)"); return (int)hipErrorInvalidValue; }
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
