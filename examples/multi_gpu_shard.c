/* examples/multi_gpu_shard.c -- ONE process, every GPU of the node: a batch of X25519 scalar multiplications held in host
 * memory, cut into contiguous shards, one host thread per device ("one process with 8 devices/streams"; the
 * records are independent units, simd/README.md:4-16 -- no exchange step, results land in one host buffer).  This is the shape
 * the reference's C callers can use: no Python, no torch.distributed, no collective -- each thread binds its device with
 * modarith_amd_set_device(), owns a stream and its device buffers, and runs
 *     upload shard -> rfc7748_X25519_batch_ws -> download shard
 * through the C ABI of include/modarith_amd.h.  The records are the reference's own: bk from the LCG of rfc7748.c:297-300
 * (rnd = 5*rnd+1 over uint16), bu = the base point u = 9 for the first half (public keys) and the previous record's key bytes
 * for the rest; the program prints an FNV-1a digest of all results, the rate, and checks record 0 of the RFC 7748 test vector
 * (rfc7748.c:271-283) on EVERY device.
 *
 *   gcc -O2 -pthread examples/multi_gpu_shard.c -Iinclude -Lmodarith_amd -l:libmodarith_amd.so \
 *       -Wl,-rpath,$PWD/modarith_amd -o examples/multi_gpu_shard && examples/multi_gpu_shard [log2 records] [devices] [out.bin] [--oversubscribe]
 * --oversubscribe: `devices` host threads whatever the number of GPUs, thread i on device i mod count -- eight shards on a
 * one-GPU box drive the library's per-device staging buffer, scratch pool and streams from eight threads at once (round 5: the
 * eight-way run before there are eight GPUs).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "modarith_amd.h"

#define Nbytes 32

typedef struct {
    int dev, rc, shard;
    size_t off, cnt;            /* the shard [off, off + cnt) */
    const char *bk, *bu;        /* page-locked host records (whole job) */
    char *bv;
    char vec[Nbytes];           /* the RFC 7748 test vector computed on this device */
    char err[256];
} shard_t;

static void from_hex(const char *src, char *dst) {
    for (int i = 0; i < Nbytes; i++) {
        unsigned v;
        sscanf(src + 2 * i, "%2x", &v);
        dst[i] = (char)v;
    }
}

#define TRY(call)                                                                     \
    do {                                                                              \
        int rc_ = (call);                                                             \
        if (rc_ != 0) {                                                               \
            snprintf(s->err, sizeof s->err, "%s: %s", #call, modarith_amd_last_error()); \
            s->rc = rc_;                                                              \
            goto out;                                                                 \
        }                                                                             \
    } while (0)

static void *run_shard(void *arg) {
    shard_t *s = (shard_t *)arg;
    void *st = NULL, *dk = NULL, *du = NULL, *dv = NULL, *ws = NULL;
    s->rc = 0;
    s->err[0] = 0;
    /* everything this thread does from here on happens on ITS device: the library keeps per-device staging and scratch */
    TRY(modarith_amd_set_device(s->dev));
    TRY(modarith_amd_stream_create(&st));
    {   /* the scalar entry point with the reference's signature works on whichever device the calling thread is bound to */
        char k[Nbytes], u[Nbytes] = {0};
        from_hex("77076d0a7318a57d3c16c17251b26645df4c2f87ebc0992ab177fba51db92c2a", k);
        u[0] = 9;
        rfc7748_X25519(k, u, s->vec);
    }
    if (s->cnt) {
        const size_t bytes = s->cnt * Nbytes, wsb = rfc7748_X25519_batch_workspace_bytes(s->cnt);
        TRY(modarith_amd_malloc(&dk, bytes));
        TRY(modarith_amd_malloc(&du, bytes));
        TRY(modarith_amd_malloc(&dv, bytes));
        TRY(modarith_amd_malloc(&ws, wsb ? wsb : 8));
        TRY(modarith_amd_memcpy_h2d(dk, s->bk + s->off * Nbytes, bytes, st));
        TRY(modarith_amd_memcpy_h2d(du, s->bu + s->off * Nbytes, bytes, st));
        TRY(rfc7748_X25519_batch_ws((const char *)dk, (const char *)du, (char *)dv, s->cnt, ws, wsb ? wsb : 8, st));
        TRY(modarith_amd_memcpy_d2h(s->bv + s->off * Nbytes, dv, bytes, st));
        TRY(modarith_amd_sync(st));
    }
out:
    if (dk) modarith_amd_free(dk);
    if (du) modarith_amd_free(du);
    if (dv) modarith_amd_free(dv);
    if (ws) modarith_amd_free(ws);
    if (st) modarith_amd_stream_destroy(st);
    return NULL;
}

int main(int argc, char **argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 16;
    const int gpus = modarith_amd_device_count();
    int oversubscribe = 0;
    for (int i = 1; i < argc; i++)
        if (!strcmp(argv[i], "--oversubscribe")) { oversubscribe = 1; for (int j = i; j + 1 < argc; j++) argv[j] = argv[j + 1]; argc--; i--; }
    int ndev = gpus;                                                             /* = number of shards = number of host threads */
    if (argc > 2 && atoi(argv[2]) > 0 && (atoi(argv[2]) < ndev || oversubscribe)) ndev = atoi(argv[2]);
    if (gpus < 1) { puts("no GPU"); return 2; }
    if (ndev > 64) ndev = 64;
    const size_t n = (size_t)1 << lg;
    void *hk, *hu, *hv;
    if (modarith_amd_host_alloc(&hk, n * Nbytes) || modarith_amd_host_alloc(&hu, n * Nbytes) || modarith_amd_host_alloc(&hv, n * Nbytes)) {
        printf("host_alloc: %s\n", modarith_amd_last_error());
        return 1;
    }
    char *bk = (char *)hk, *bu = (char *)hu, *bv = (char *)hv;
    uint16_t rnd = 1;
    for (size_t j = 0; j < n; j++)
        for (int i = 0; i < Nbytes; i++) {
            rnd = (uint16_t)(5 * rnd + 1);
            bk[j * Nbytes + i] = (char)(rnd % 256);
        }
    for (size_t j = 0; j < n; j++) {
        if (j < n / 2 || j == 0) { memset(bu + j * Nbytes, 0, Nbytes); bu[j * Nbytes] = 9; }
        else memcpy(bu + j * Nbytes, bk + (j - 1) * Nbytes, Nbytes);          /* arbitrary 256-bit u (modimp reduces it) */
    }

    shard_t sh[64];
    pthread_t th[64];
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int d = 0; d < ndev; d++) {
        sh[d].dev = d % gpus;
        sh[d].shard = d;
        sh[d].off = n * (size_t)d / (size_t)ndev;                              /* contiguous index blocks */
        sh[d].cnt = n * (size_t)(d + 1) / (size_t)ndev - sh[d].off;
        sh[d].bk = bk; sh[d].bu = bu; sh[d].bv = bv;
        pthread_create(&th[d], NULL, run_shard, &sh[d]);
    }
    for (int d = 0; d < ndev; d++) pthread_join(th[d], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double dt = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);

    int bad = 0;
    char want[Nbytes];
    from_hex("8520f0098930a754748b7ddcb43ef75a0dbf3a0d26381af4eba4a98eaa9b4e6a", want);
    for (int d = 0; d < ndev; d++) {
        if (sh[d].rc) { printf("shard %d on device %d: %s\n", d, sh[d].dev, sh[d].err); bad = 1; }
        else if (memcmp(sh[d].vec, want, Nbytes)) { printf("shard %d on device %d: RFC 7748 test vector differs\n", d, sh[d].dev); bad = 1; }
        else printf("shard %d on device %d: records [%zu, %zu) ok, RFC 7748 vector ok\n", d, sh[d].dev, sh[d].off, sh[d].off + sh[d].cnt);
    }
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n * Nbytes; i++) { h ^= (unsigned char)bv[i]; h *= 0x100000001b3ull; }
    if (modarith_amd_status() != 0) { printf("a scalar entry point recorded error %d: %s\n", modarith_amd_status(), modarith_amd_last_error()); bad = 1; }
    printf("devices %d records %zu seconds %.3f rate %.3e per s (first call on each device included) digest %016llx\n", ndev < gpus ? ndev : gpus, n, dt, (double)n / dt,
           (unsigned long long)h);
    if (oversubscribe) printf("oversubscribed: %d shards on %d device(s)\n", ndev, gpus);
    if (argc > 3) {                                                              /* the bytes, for a checker */
        FILE *f = fopen(argv[3], "wb");
        if (!f || fwrite(bk, 1, n * Nbytes, f) != n * Nbytes || fwrite(bu, 1, n * Nbytes, f) != n * Nbytes || fwrite(bv, 1, n * Nbytes, f) != n * Nbytes) bad = 1;
        if (f) fclose(f);
    }
    modarith_amd_host_free(hk); modarith_amd_host_free(hu); modarith_amd_host_free(hv);
    return bad;
}
