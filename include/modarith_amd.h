/* include/modarith_amd.h -- C ABI of libmodarith_amd.so, the MI355X (gfx950) batched finite-field engine.
 *
 * This is the drop-in boundary for modarith's generated field arithmetic.  The reference has no
 * linked library: consumers paste field.c at a marker (rfc7748.c:24-28, edwards.c:19-23, edge.c:5-9).
 * Its only linked form is the transient test.so that the generators build with makestatic=False and
 * drive through ctypes (pseudo.py:1689-1750, monty.py:2264-2325); those exported signatures are what
 * this header reproduces, in two forms per function and prime:
 *
 *   <fn>_<PRIME>_ct(...)      scalar form, the reference signature verbatim with the generators'
 *                             decorated name (decoration=True, pseudo.py:1940-1944).  Host pointers,
 *                             one element; executed on the GPU (n = 1).  For tests and glue only.
 *   <fn>_<PRIME>_batch(...)   batched form, the n-lane generalisation the reference itself uses for
 *                             SIMD/SIMT (simd/pseudo_simd.py: same names and arity, spint widened to a
 *                             lane vector, limb-major storage; simd/rfc7748_simt.cu:165-168 contiguous
 *                             byte records).  DEVICE pointers.
 *
 * PRIME is one of X25519 (pseudo.py 64 X25519: 5 x 51-bit limbs), NIST256 (monty.py 64 NIST256:
 * 5 x 52-bit, Montgomery form, R = 2^260), X448 (monty.py 64 X448: 8 x 56-bit, Montgomery form,
 * R = 2^504).  spint = uint64_t as in the 64-bit field.c (pseudo.py:1394-1398).
 *
 * Batched layout: limb-interleaved SoA, in one of two forms selected by the limb stride ld (in elements):
 *   FLAT  (ld >= n): buf[limb*ld + j], 0 <= limb < Nlimbs, 0 <= j < n ("lanes" of simd.h, j-major within a limb).  ld lets
 *         a call work on a slice [off, off+n) of a larger batch (pass buf+off).
 *   TILED (ld < n, ld a power of two >= 128): the batch is a sequence of tiles of ld elements, each tile limb-interleaved with
 *         stride ld and Nlimbs*ld words long: buf[((j / ld)*Nlimbs + limb)*ld + (j % ld)]; the buffer holds ceil(n/ld) whole
 *         tiles.  All operands of one call share ld.  RECOMMENDED for large batches, ld = 4096: the Nlimbs rows a workgroup
 *         streams then lie in one contiguous 160-256 KiB stretch instead of Nlimbs stretches n*8 bytes apart, and the HBM rate
 *         stops depending on where the driver placed the arrays (+6 % on the slow placements of 5-limb fields, +11-16 % for
 *         8-limb fields).  Every per-prime field function and the AoS converters take both forms; the
 *         curve API (ecn_*) and the byte-record functions' record side are flat / AoS only.
 * For the 16-byte fast path keep buffers 16-byte aligned and ld even.  Byte records
 * (modimp/modexp/rfc7748) are contiguous AoS: rec[j*Nbytes + i].
 * Ownership / aliasing: the caller owns every buffer, nothing is allocated or retained; an output
 * may be the same buffer (same base, same ld) as an input, as in the reference (modsqr(a,a),
 * modadd(z,z,z), modmul(z2,E,z2), rfc7748(alice,apk,apk)).  Partial overlap is not supported.
 * Results are bit-identical to the reference's field.c, including non-canonical (< 2p) limbs;
 * modpro/modinv use a different addition chain (the reference shells out to `addchain`), so their
 * limbs are only comparable after redc.  modinv (both forms, with or without a progenitor) returns the inverse in NORMALISED
 * form nres(redc(1/x)): words that depend on the value alone.  modinv_<P>_batch shares one inversion between up to 64
 * elements of a large batch (Montgomery's simultaneous inversion: 3 multiplications per element instead of ~265; zero
 * values are kept out of the shared product by lane predication, modinv(0) = 0) -- the same words as one inversion per
 * element, 20-30 x the rate.
 * Errors: the reference signals none.  Batched calls return 0 or a hipError_t value (launch/device
 * errors only); modarith_amd_last_error() describes the last failure on the calling thread.  Scalar
 * _ct calls keep the reference's void / predicate signatures and therefore cannot return a device error: they RECORD it
 * (modarith_amd_last_error() on the calling thread, and the process-wide sticky modarith_amd_status()), launch nothing and
 * hand back zero-filled outputs.  Nothing in the library calls abort() or exit().
 * Streams: `stream` is a hipStream_t (NULL = default stream); calls are asynchronous on it.
 * Threading: entry points are re-entrant; scalar _ct calls serialise on the staging buffer of THEIR device (one buffer and
 * one mutex per device: scalar calls on different devices run side by side).
 */
#ifndef MODARITH_AMD_H
#define MODARITH_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t ma_spint; /* spint of the 64-bit field.c (pseudo.py:1395) */

/* ---- library / device utilities (no counterpart in the reference: it has no runtime) ---- */
int modarith_amd_abi_version(void);            /* = MODARITH_AMD_ABI */
const char *modarith_amd_last_error(void);
/* the first error code (a hipError_t value) any scalar _ct entry point of this process has recorded since the last clear; 0 = none.
 * A caller of the reference's void signatures polls this where field.c's caller would have had nothing to check. */
int modarith_amd_status(void);
void modarith_amd_clear_status(void);
/* the same for the CALLING THREAD alone: the first error its own scalar calls have recorded since its last clear (a thread that shares
 * the process with others checks this one).  A failed scalar call launches nothing; it zero-fills its pure outputs, leaves operands
 * that are also inputs (modnsqr(a, n), modmul(z, e, z), ecn dbl) as they were, and its predicates (modis0 modis1 modsign modcmp modqr
 * modimp, ecn cmp / isinf / get) answer -1 instead of the reference's 0 / 1. */
int modarith_amd_thread_status(void);
void modarith_amd_clear_thread_status(void);
int modarith_amd_device_count(void);
int modarith_amd_set_device(int dev);
int modarith_amd_malloc(void **dptr, size_t bytes);
int modarith_amd_free(void *dptr);
int modarith_amd_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream);
int modarith_amd_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream);
int modarith_amd_sync(void *stream);
/* streams and page-locked host memory, so that a plain-C caller holding its data on the host can overlap the
 * upload of chunk i+1, the kernels of chunk i and the download of chunk i-1:
 * stream_wait makes `stream` wait for everything enqueued on `other` so far (an event underneath) */
int modarith_amd_stream_create(void **stream);
int modarith_amd_stream_destroy(void *stream);
int modarith_amd_stream_wait(void *stream, void *other);
int modarith_amd_host_alloc(void **hptr, size_t bytes);
int modarith_amd_host_free(void *hptr);
/* element-major (spint x[n][nlimbs], how CPU callers hold arrays of elements) <-> SoA (flat or tiled, by ld as above); any limb count */
int modarith_amd_aos_to_soa(const ma_spint *aos, ma_spint *soa, size_t n, int nlimbs, size_t ld, void *stream);
int modarith_amd_soa_to_aos(const ma_spint *soa, ma_spint *aos, size_t n, int nlimbs, size_t ld, void *stream);
/* layout helpers: the limb stride a caller without a layout of its own should use for a batch of n elements (tiles of 4096 once
 * the batch holds two of them, else flat rows: the TILED note above), and the number of 64-bit words a batch of n elements of
 * nlimbs limbs occupies with stride ld (flat: nlimbs*ld; tiled: ceil(n/ld) whole tiles) */
size_t modarith_amd_recommended_ld(size_t n);
/* ... with the shape of the field (round 6, profiles/r06_tile_shape_sweep.log): 8192 for elements of eight or more limbs once the batch
 * holds two such tiles, else as above */
size_t modarith_amd_recommended_ld_for(size_t n, int nlimbs);
size_t modarith_amd_batch_words(size_t n, int nlimbs, size_t ld);
/* the library's stream-ordered scratch (the split form of rfc7748_<C>_batch, in-place modinv_<P>_batch) caches up to 1.25 GiB per
 * device between calls; scratch_trim gives what is cached beyond keep_bytes on the current device back to the driver */
int modarith_amd_scratch_trim(size_t keep_bytes);
/* diagnostic: the shader clock a kernel sees WHILE other work runs.  Enqueue on a second stream in front of the work to be observed:
 * one wave waits delay_us of wall clock, then reads the shader-clock counter (s_memtime) and the constant-rate wall clock
 * (s_memrealtime) at both ends of a window of window_us and stores the two differences in out[0] (shader cycles) and out[1]
 * (wall-clock ticks), device memory.  clock = out[0] / out[1] x modarith_amd_wall_clock_khz() kHz.  bench.py brackets every
 * VALU-bound leg with it, so that a rate can be told from the clock the part held while it was measured. */
int modarith_amd_sclk_probe(uint64_t *out, unsigned delay_us, unsigned window_us, void *stream);
int modarith_amd_wall_clock_khz(void);
/* diagnostic: the name of the kernel family this thread's last batched call launched ("rfc7748(split)", "rfc7748(field form)", ...) */
const char *modarith_amd_last_launch(void);
/* per-prime macro block of field.c (pseudo.py:1403-1407): returns 0 if `prime` is unknown */
int modarith_amd_field_info(const char *prime, int *nlimbs, int *radix, int *nbits, int *nbytes, int *montgomery);

/* ABI history.  No entry point has changed its signature or meaning since ABI 1; the number moves when a caller built against an older
 * header could see different behaviour from an unchanged call:
 *   1  rounds 1-5.
 *   2  round 6.  (a) *_workspace_bytes(n) values differ from earlier builds (round 5: the P-256 fused forms grew to 0.67-1.5 GB for
 *      their per-record window tables; round 6: ecn_ed25519 / ed448_mul2_get add 127 bytes of alignment slack): ask the function of
 *      THIS library before every allocation, never keep a size across library versions.  (b) A caller workspace may sit at any
 *      address; the library rounds it up itself (round 5 fell back to its own pool without a word when ecn_*_mul2_get got an
 *      unaligned one).  (c) prop_<PRIME>_ct / _batch added.  (d) A failed scalar call leaves in/out operands untouched (see
 *      modarith_amd_status). */
#define MODARITH_AMD_ABI 2

/* One block of declarations per prime.  Reference emitters cited once here:
 *   prop     pseudo.py:223-251 / monty.py:352-380
 *   flatten  pseudo.py:255-269    modfsb pseudo.py:272-283     modadd pseudo.py:286-304
 *   modsub   pseudo.py:307-326    modneg pseudo.py:329-348     modmli pseudo.py:705-728 / monty.py:876-978
 *   modmul   pseudo.py:616-659 / monty.py:663-872              modsqr pseudo.py:663-702 / monty.py:982-1165
 *   modcpy   pseudo.py:730-743    modnsqr pseudo.py:745-755    modpro pseudo.py:758-785
 *   modqr    pseudo.py:815-831    modsqrt pseudo.py:834-874
 *   modinv   pseudo.py:788-812    nres pseudo.py:952-962 / monty.py:1386-1399
 *   redc     pseudo.py:965-976 / monty.py:1402-1416            modis1 pseudo.py:877-891
 *   modis0   pseudo.py:894-906    modzer 909-919  modone 922-934  modint 937-949
 *   modcmv   pseudo.py:1017-1048  modcsw pseudo.py:979-1014    modshl 1052-1065  modshr 1068-1081
 *   modhaf   pseudo.py:1084-1100  mod2r 1102-1112  modexp 1115-1127  modimp 1130-1146
 *   modsign  pseudo.py:1149-1158  modcmp 1161-1174
 *   generic=False modadd/modsub/modneg ("_lazy"): pseudo.py:294-302, 315-324, 337-346 with mp=2
 *
 * The same macro declares the entry points of a field made by the generator mode (python -m modarith_amd.generate 64 <prime>,
 * the counterpart of running pseudo.py / monty.py on a prime of one's own): MODARITH_AMD_DECLARE(2519) for 2^251-9, whose
 * definitions live in modarith_amd/plugins/libmodarith_amd_2519.so (link it next to libmodarith_amd.so).
 */
#define MODARITH_AMD_DECLARE(P)                                                                                         \
    /* ---------------- scalar form: reference signatures, host pointers ---------------- */                            \
    ma_spint prop_##P##_ct(ma_spint *n);        /* static in field.c; exported so that all 32 emitted names exist */    \
    ma_spint flatten_##P##_ct(ma_spint *n);                                                                             \
    ma_spint modfsb_##P##_ct(ma_spint *n);                                                                              \
    void modadd_##P##_ct(const ma_spint *a, const ma_spint *b, ma_spint *n);                                            \
    void modsub_##P##_ct(const ma_spint *a, const ma_spint *b, ma_spint *n);                                            \
    void modneg_##P##_ct(const ma_spint *b, ma_spint *n);                                                               \
    void modmli_##P##_ct(const ma_spint *a, int b, ma_spint *c);                                                        \
    void modmul_##P##_ct(const ma_spint *a, const ma_spint *b, ma_spint *c);                                            \
    void modsqr_##P##_ct(const ma_spint *a, ma_spint *c);                                                               \
    void modcpy_##P##_ct(const ma_spint *a, ma_spint *c);                                                               \
    void modnsqr_##P##_ct(ma_spint *a, int n);                                                                          \
    void modpro_##P##_ct(const ma_spint *w, ma_spint *z);                                                               \
    void modinv_##P##_ct(const ma_spint *x, const ma_spint *h, ma_spint *z); /* h may be NULL */                        \
    int modqr_##P##_ct(const ma_spint *h, const ma_spint *x);                /* h may be NULL */                        \
    void modsqrt_##P##_ct(const ma_spint *x, const ma_spint *h, ma_spint *r); /* h may be NULL */                       \
    void nres_##P##_ct(const ma_spint *m, ma_spint *n);                                                                 \
    void redc_##P##_ct(const ma_spint *n, ma_spint *m);                                                                 \
    int modis1_##P##_ct(const ma_spint *a);                                                                             \
    int modis0_##P##_ct(const ma_spint *a);                                                                             \
    void modzer_##P##_ct(ma_spint *a);                                                                                  \
    void modone_##P##_ct(ma_spint *a);                                                                                  \
    void modint_##P##_ct(int x, ma_spint *a);                                                                           \
    void modcmv_##P##_ct(int b, const ma_spint *g, volatile ma_spint *f);                                               \
    void modcsw_##P##_ct(int b, volatile ma_spint *g, volatile ma_spint *f);                                            \
    void modshl_##P##_ct(unsigned int n, ma_spint *a);                                                                  \
    int modshr_##P##_ct(unsigned int n, ma_spint *a);                                                                   \
    void modhaf_##P##_ct(ma_spint *n);                                                                                  \
    void mod2r_##P##_ct(unsigned int r, ma_spint *a);                                                                   \
    void modexp_##P##_ct(const ma_spint *a, char *b);                                                                   \
    int modimp_##P##_ct(const char *b, ma_spint *a);                                                                    \
    int modsign_##P##_ct(const ma_spint *a);                                                                            \
    int modcmp_##P##_ct(const ma_spint *a, const ma_spint *b);                                                          \
    /* ---------------- batched form: device pointers, SoA, limb stride ld ---------------- */                          \
    int modadd_##P##_batch(const ma_spint *a, const ma_spint *b, ma_spint *n_, size_t n, size_t ld, void *stream);      \
    int modsub_##P##_batch(const ma_spint *a, const ma_spint *b, ma_spint *n_, size_t n, size_t ld, void *stream);      \
    int modneg_##P##_batch(const ma_spint *b, ma_spint *n_, size_t n, size_t ld, void *stream);                         \
    int modadd_lazy_##P##_batch(const ma_spint *a, const ma_spint *b, ma_spint *n_, size_t n, size_t ld, void *stream); \
    int modsub_lazy_##P##_batch(const ma_spint *a, const ma_spint *b, ma_spint *n_, size_t n, size_t ld, void *stream); \
    int modneg_lazy_##P##_batch(const ma_spint *b, ma_spint *n_, size_t n, size_t ld, void *stream);                    \
    int modmul_##P##_batch(const ma_spint *a, const ma_spint *b, ma_spint *c, size_t n, size_t ld, void *stream);       \
    /* shared multiplicand c[j] = a[j] * b0; b0 = Nlimbs limbs in HOST memory (nres/redc/curve-constant call sites) */  \
    int modmuls_##P##_batch(const ma_spint *a, const ma_spint *b0_host, ma_spint *c, size_t n, size_t ld, void *stream);\
    int modsqr_##P##_batch(const ma_spint *a, ma_spint *c, size_t n, size_t ld, void *stream);                          \
    int modmli_##P##_batch(const ma_spint *a, int b, ma_spint *c, size_t n, size_t ld, void *stream);                   \
    int modcpy_##P##_batch(const ma_spint *a, ma_spint *c, size_t n, size_t ld, void *stream);                          \
    int modnsqr_##P##_batch(ma_spint *a, int k, size_t n, size_t ld, void *stream);                                     \
    int modpro_##P##_batch(const ma_spint *w, ma_spint *z, size_t n, size_t ld, void *stream);                          \
    int modinv_##P##_batch(const ma_spint *x, const ma_spint *h, ma_spint *z, size_t n, size_t ld, void *stream);       \
    int modsqrt_##P##_batch(const ma_spint *x, const ma_spint *h, ma_spint *r, size_t n, size_t ld, void *stream);      \
    int modqr_##P##_batch(const ma_spint *h, const ma_spint *x, int *out, size_t n, size_t ld, void *stream);           \
    int nres_##P##_batch(const ma_spint *m, ma_spint *n_, size_t n, size_t ld, void *stream);                           \
    int redc_##P##_batch(const ma_spint *n_, ma_spint *m, size_t n, size_t ld, void *stream);                           \
    /* in place; flag (device int[n], may be NULL) receives the return value per element */                            \
    int modfsb_##P##_batch(ma_spint *a, int *flag, size_t n, size_t ld, void *stream);                                  \
    int flatten_##P##_batch(ma_spint *a, int *flag, size_t n, size_t ld, void *stream);                                 \
    int prop_##P##_batch(ma_spint *a, int *flag, size_t n, size_t ld, void *stream);     /* flag: -1 / 0 (the mask) */   \
    int modhaf_##P##_batch(ma_spint *a, size_t n, size_t ld, void *stream);                                             \
    int modshl_##P##_batch(unsigned int k, ma_spint *a, size_t n, size_t ld, void *stream);                             \
    int modshr_##P##_batch(unsigned int k, ma_spint *a, int *out, size_t n, size_t ld, void *stream);                   \
    /* predicates: out = device int[n] */                                                                               \
    int modis1_##P##_batch(const ma_spint *a, int *out, size_t n, size_t ld, void *stream);                             \
    int modis0_##P##_batch(const ma_spint *a, int *out, size_t n, size_t ld, void *stream);                             \
    int modsign_##P##_batch(const ma_spint *a, int *out, size_t n, size_t ld, void *stream);                            \
    /* not in field.c: out[j] = 1 when every limb of element j is below 2^(Radix+2), the limb budget the generators'  */ \
    /* functions keep (SURVEY 8b "Errors") and the curve-layer kernels rely on; 0 flags a fabricated operand             */ \
    int modlimbs_##P##_batch(const ma_spint *a, int *out, size_t n, size_t ld, void *stream);                           \
    int modcmp_##P##_batch(const ma_spint *a, const ma_spint *b, int *out, size_t n, size_t ld, void *stream);          \
    /* fills */                                                                                                         \
    int modzer_##P##_batch(ma_spint *a, size_t n, size_t ld, void *stream);                                             \
    int modone_##P##_batch(ma_spint *a, size_t n, size_t ld, void *stream);                                             \
    int modint_##P##_batch(int x, ma_spint *a, size_t n, size_t ld, void *stream);                                      \
    int mod2r_##P##_batch(unsigned int r, ma_spint *a, size_t n, size_t ld, void *stream);                              \
    /* constant-time conditional move/swap, one selector d[j] in {0,1} per element (device int[n]) */                  \
    int modcmv_##P##_batch(const int *d, const ma_spint *g, ma_spint *f, size_t n, size_t ld, void *stream);            \
    int modcsw_##P##_batch(const int *d, ma_spint *g, ma_spint *f, size_t n, size_t ld, void *stream);                  \
    /* the reference's timing protocol (time.c: pseudo.py:1177-1386; CUDA form simd/pseudo_cuda.py:1163-1231) run per  \
       lane in registers: kind 0 = outer*200*5 dependent modmul on (x,y), 1 = outer*500*2 modsqr on x, 2 = outer*2     \
       modinv on x; inputs are plain (not nres'd) limbs as time.c bakes them in; z[j] = redc(result), so              \
       z[0][j] & 0xFFFFFF is the reference's check word */                                                            \
    int time_protocol_##P##_batch(int kind, const ma_spint *x, const ma_spint *y, ma_spint *z, long outer, size_t n,    \
                                  size_t ld, void *stream);                                                             \
    /* synthetic inputs (SURVEY 8(d) recipe; not a reference function): out[j] = canonical limbs of a value uniform in  \
       [0,p) -- the splitmix64 stream keyed by (seed, array) read at element first+j, ceil(Nbits/64)+1 words reduced     \
       mod p; plus_p != 0 adds p and leaves the top limb unmasked (a representative in [p,2p)).  Plain values: apply     \
       nres for Montgomery form.  Host model: tests/util.py uniform_model */                                            \
    int moduniform_##P##_batch(unsigned long long seed, unsigned long long array, size_t first, int plus_p,             \
                               ma_spint *out, size_t n, size_t ld, void *stream);                                       \
    /* byte records: device char[n*Nbytes], big-endian per record as modimp/modexp take them */                        \
    int modimp_##P##_batch(const char *b, ma_spint *a, int *flag, size_t n, size_t ld, void *stream);                   \
    int modexp_##P##_batch(const ma_spint *a, char *b, size_t n, size_t ld, void *stream);

MODARITH_AMD_DECLARE(X25519)
MODARITH_AMD_DECLARE(NIST256)
MODARITH_AMD_DECLARE(X448)
/* further primes of the generators' named lists (pseudo.py:1498-1548, monty.py:1966-2062) and the group orders
 * curve.py:324-329 feeds to monty.py; same functions, constants from modarith_amd/params.py:
 *   pseudo-Mersenne: NIST521 (9 x 58), PM266 (5 x 54), PM383 (7 x 55, non-EPM rows), NUMS256W (5 x 52)
 *   Montgomery     : NIST384 (7 x 56), NIST224 (4 x 56 + virtual limb), SECP256K1M (5 x 52), all full reduction
 *                    (ndash != 1); group orders NIST256Q, ED25519Q, ED448Q (general primes);
 *                    GM270 GM240 GM360 GM480 GM384 GM512 (generalised Mersenne trinomials), TWEEDLE, SIDH434, SIDH503
 *   pseudo-Mersenne: also C2065 (4 x 52), PM336 (6 x 56), PM512 (9 x 57, non-EPM rows), and the split-high-part
 *                    ("overflow") rows of pseudo.py:1640-1657: SECP256K1 (5 x 52, the field curve.py gives secp256k1 at
 *                    64 bits) and C41417 (7 x 60).  SECP256K1M is monty.py's flavour of the secp256k1 prime.
 *   Montgomery     : also the fields of curve.py's ED248, ED376, ED500 (5*2^248-1, 65*2^376-1, 27*2^500-1) */
MODARITH_AMD_DECLARE(NIST521)
MODARITH_AMD_DECLARE(PM266)
MODARITH_AMD_DECLARE(PM383)
MODARITH_AMD_DECLARE(NUMS256W)
MODARITH_AMD_DECLARE(NIST384)
MODARITH_AMD_DECLARE(NIST224)
MODARITH_AMD_DECLARE(SECP256K1M)
MODARITH_AMD_DECLARE(NIST256Q)
MODARITH_AMD_DECLARE(ED25519Q)
MODARITH_AMD_DECLARE(ED448Q)
MODARITH_AMD_DECLARE(C2065)
MODARITH_AMD_DECLARE(PM336)
MODARITH_AMD_DECLARE(PM512)
MODARITH_AMD_DECLARE(GM270)
MODARITH_AMD_DECLARE(GM240)
MODARITH_AMD_DECLARE(GM360)
MODARITH_AMD_DECLARE(GM480)
MODARITH_AMD_DECLARE(GM384)
MODARITH_AMD_DECLARE(GM512)
MODARITH_AMD_DECLARE(TWEEDLE)
MODARITH_AMD_DECLARE(SIDH434)
MODARITH_AMD_DECLARE(SIDH503)
MODARITH_AMD_DECLARE(SECP256K1)
MODARITH_AMD_DECLARE(C41417)
MODARITH_AMD_DECLARE(ED248)
MODARITH_AMD_DECLARE(ED376)
MODARITH_AMD_DECLARE(ED500)
MODARITH_AMD_DECLARE(SIDH610)
MODARITH_AMD_DECLARE(SIDH751)
MODARITH_AMD_DECLARE(MFP4)
MODARITH_AMD_DECLARE(MFP7)
MODARITH_AMD_DECLARE(MFP1973)
MODARITH_AMD_DECLARE(CSIDH512)
MODARITH_AMD_DECLARE(GM378)
MODARITH_AMD_DECLARE(PM383M)
MODARITH_AMD_DECLARE(PM266M)
MODARITH_AMD_DECLARE(PM336M)
MODARITH_AMD_DECLARE(C41417M)
MODARITH_AMD_DECLARE(PM512M)
MODARITH_AMD_DECLARE(M607)

/* RFC 7748 ladder, bv = [bk] * bu (reference rfc7748.c:156 `void rfc7748(const char *bk,const char *bu,char *bv)`).
 * Scalar form: host pointers, Nbytes each (32 / 56), RFC little-endian.  Batched form: device pointers,
 * n contiguous records (simd/rfc7748_simt.cu:165-168,224); bv may alias bu. */
void rfc7748_X25519(const char *bk, const char *bu, char *bv);
void rfc7748_X448(const char *bk, const char *bu, char *bv);
int rfc7748_X25519_batch(const char *bk, const char *bu, char *bv, size_t n, void *stream);
int rfc7748_X448_batch(const char *bk, const char *bu, char *bv, size_t n, void *stream);
/* The batched form does not owe one field inversion (rfc7748.c:225-254 modinv: 7-8 % of the function) to every record: for
 * n >= 8192 it runs the ladders in one kernel and finishes up to 32 records per inversion in a second one (Montgomery's
 * simultaneous inversion; z2 = 0 handled by lane predication) -- the same bytes for every input.  rfc7748_<C>_batch takes
 * the scratch this needs (rfc7748_<C>_batch_workspace_bytes(n): Nbytes + 40 / 64 bytes per record) from a stream-ordered
 * pool of the library's own, and falls back to one inversion per record while the stream is being captured or when the
 * pool is unavailable; rfc7748_<C>_batch_ws takes it from the caller (device memory, 8-byte aligned, any n) and is what a
 * resident caller or a graph capture uses. */
size_t rfc7748_X25519_batch_workspace_bytes(size_t n);
size_t rfc7748_X448_batch_workspace_bytes(size_t n);
int rfc7748_X25519_batch_ws(const char *bk, const char *bu, char *bv, size_t n, void *workspace, size_t workspace_bytes, void *stream);
int rfc7748_X448_batch_ws(const char *bk, const char *bu, char *bv, size_t n, void *workspace, size_t workspace_bytes, void *stream);
/* rfc7748() on the curve's BASE POINT (u = 9 / u = 5): public-key generation, the first half of every exchange in the
 * reference's main() (rfc7748.c:297-333 `rfc7748(alice, base, apk)`).  Same bytes as rfc7748_<C>_batch with bu = the base
 * point; computed on the birationally equivalent / 4-isogenous Edwards curve from a fixed-base table (no ladder steps). */
int rfc7748_X25519_base_batch(const char *bk, char *bv, size_t n, void *stream);
int rfc7748_X448_base_batch(const char *bk, char *bv, size_t n, void *stream);

/* ---- Curve layer on the field path: the API of curve.h:13-29 with XXX = _<curve>_
 * (curve.py:344-345), for ED25519 (over the X25519 field), ED448 (over the X448 field), and NUMS256E (over 2^256-189), ED248,
 * ED376, ED500 with CONSTANT_B and CONSTANT_X kept as C ints.
 * A point is projective (x:y:z), `struct xyz` of curve.py:304-309.  Scalar form: host `point`, one element
 * on the GPU.  Batched form: device SoA P[(c*Nlimbs + i)*ld + j], c = 0,1,2 for x,y,z -- a host point is
 * that layout with ld = 1.  Scalars e and coordinates x,y are big-endian Nbytes records, as in the
 * reference.  ecn_*_mul is the constant-time 4-bit fixed-window multiplication (edwards.c:435-482); its
 * 9-entry table (two of them for mul2) lives in a caller-provided device workspace of
 * ecn_*_mul_workspace_bytes(n) bytes: per-wave slabs for the RESIDENT grid (64 lanes x waves per SIMD x 1024 SIMDs), not for
 * n, so the size stops growing at the grid -- 283 MB for 5-limb fields at two waves per SIMD, up to 566 MB for ed25519, whose
 * kernels run at four waves per SIMD since round 4 (mul alone touches half of it, mul2 all of it; more than the 256 MiB
 * Infinity Cache: the scans are whole 512-byte rows from HBM / L2).  Ask the function, do not hard-code a size.
 * add, dbl and mul run the reference's formulas (edwards.c:73-145, weierstrass.c:68-281) from the bit-exact field calls
 * in the reference's order, and their PROJECTIVE LIMBS equal the reference's: the reference's own edwards.c / weierstrass.c,
 * built in the build container from its files without the functions that need the external addchain tool, produced the
 * fixtures tests/golden/curveref_<CURVE>.json (gen, mul, dbl, add, sub, neg, cof, the special cases; eleven curves), which the
 * kernels and the oracle reproduce limb for limb.  set/get/affine involve modpro and are comparable as affine coordinates
 * (big-integer model, survey-captured reference outputs).  ecnXXXmul2 (a joint sparse form
 * with data-dependent branches in the reference, edwards.c:404-431, 486-510) is two interleaved fixed-window
 * multiplications sharing their doublings here, constant-time: same point, another projective representative;
 * ecn_*_mul2_exact_batch walks the reference's joint sparse form itself and returns the reference's limbs (variable time);
 * the scalar ecn_*_mul2 (one element, nothing to keep in step) takes that form.
 * Input points must have limbs below 2^(Radix+2) and coordinates that are field elements in the API's sense (any representative
 * below 2p) -- true of every point these functions or the reference's produce; the field-level functions above have no such
 * condition.  (The scalar multiplications of ed25519 hold their elements in half-limb form, csrc/fh51.h: same limbs for such inputs.) */
#define MODARITH_AMD_DECLARE_EDWARDS(c, NL)                                                                             \
    typedef struct { ma_spint x[NL], y[NL], z[NL]; } ma_point_##c##_t;                                                  \
    int ecn_##c##_get(ma_point_##c##_t *P, char *x, char *y);                          /* edwards.c:221-239 */        \
    void ecn_##c##_set(int s, const char *x, const char *y, ma_point_##c##_t *P);      /* edwards.c:347-366 */        \
    void ecn_##c##_inf(ma_point_##c##_t *P);                                                                            \
    int ecn_##c##_isinf(ma_point_##c##_t *P);                                                                           \
    void ecn_##c##_neg(ma_point_##c##_t *P);                                                                            \
    void ecn_##c##_add(ma_point_##c##_t *Q, ma_point_##c##_t *P);                      /* P += Q, edwards.c:73-111 */ \
    void ecn_##c##_sub(ma_point_##c##_t *Q, ma_point_##c##_t *P);                                                       \
    void ecn_##c##_dbl(ma_point_##c##_t *P);                                           /* edwards.c:123-145 */        \
    void ecn_##c##_gen(ma_point_##c##_t *P);                                                                            \
    void ecn_##c##_mul(const char *e, ma_point_##c##_t *P);                            /* edwards.c:435-482 */        \
    void ecn_##c##_mul2(const char *e, ma_point_##c##_t *P, const char *f, ma_point_##c##_t *Q, ma_point_##c##_t *R);  \
    void ecn_##c##_ran(int r, ma_point_##c##_t *P);                                    /* edwards.c:55-63 */          \
    int ecn_##c##_cmp(ma_point_##c##_t *P, ma_point_##c##_t *Q);                                                        \
    void ecn_##c##_affine(ma_point_##c##_t *P);                                                                         \
    void ecn_##c##_cpy(ma_point_##c##_t *Q, ma_point_##c##_t *P);                                                       \
    void ecn_##c##_cof(ma_point_##c##_t *P);                                                                            \
    size_t ecn_##c##_mul_workspace_bytes(size_t n);                                                                     \
    int ecn_##c##_mul_batch(const char *e, ma_spint *P, size_t n, size_t ld, void *workspace, size_t workspace_bytes,   \
                            void *stream);                                                                              \
    /* R = eP + fQ (edwards.c:486-510); same workspace size as mul; R may be P or Q */                                 \
    int ecn_##c##_mul2_batch(const char *e, const ma_spint *P, const char *f, const ma_spint *Q, ma_spint *R, size_t n, \
                             size_t ld, void *workspace, size_t workspace_bytes, void *stream);                         \
    /* the same R = eP + fQ by the reference's own walk over its joint sparse form (variable time, as the reference's):    \
       the reference's projective limbs, where mul2_batch gives another representative of the same point */           \
    int ecn_##c##_mul2_exact_batch(const char *e, const ma_spint *P, const char *f, const ma_spint *Q, ma_spint *R,    \
                                   size_t n, size_t ld, void *workspace, size_t workspace_bytes, void *stream);         \
    int ecn_##c##_ran_batch(int r, ma_spint *P, size_t n, size_t ld, void *stream);                                     \
    int ecn_##c##_add_batch(const ma_spint *Q, ma_spint *P, size_t n, size_t ld, void *stream);                         \
    int ecn_##c##_sub_batch(const ma_spint *Q, ma_spint *P, size_t n, size_t ld, void *stream);                         \
    int ecn_##c##_cpy_batch(const ma_spint *Q, ma_spint *P, size_t n, size_t ld, void *stream);                         \
    int ecn_##c##_dbl_batch(ma_spint *P, size_t n, size_t ld, void *stream);                                            \
    int ecn_##c##_neg_batch(ma_spint *P, size_t n, size_t ld, void *stream);                                            \
    int ecn_##c##_inf_batch(ma_spint *P, size_t n, size_t ld, void *stream);                                            \
    int ecn_##c##_gen_batch(ma_spint *P, size_t n, size_t ld, void *stream);                                            \
    int ecn_##c##_cof_batch(ma_spint *P, size_t n, size_t ld, void *stream);                                            \
    int ecn_##c##_affine_batch(ma_spint *P, size_t n, size_t ld, void *stream);                                         \
    int ecn_##c##_cmp_batch(const ma_spint *P, const ma_spint *Q, int *out, size_t n, size_t ld, void *stream);         \
    int ecn_##c##_isinf_batch(const ma_spint *P, int *out, size_t n, size_t ld, void *stream);                          \
    /* s: device int[n] of sign bits or NULL (= 0); x, y: device byte records, either may be NULL */                   \
    int ecn_##c##_set_batch(const int *s, const char *x, const char *y, ma_spint *P, size_t n, size_t ld, void *stream);\
    /* makes P affine in place; x, y, sign (device) may each be NULL */                                                \
    int ecn_##c##_get_batch(ma_spint *P, char *x, char *y, int *sign, size_t n, size_t ld, void *stream);

MODARITH_AMD_DECLARE_EDWARDS(ed25519, 5)
MODARITH_AMD_DECLARE_EDWARDS(ed448, 8)
MODARITH_AMD_DECLARE_EDWARDS(nums256e, 5)
MODARITH_AMD_DECLARE_EDWARDS(ed248, 5)
MODARITH_AMD_DECLARE_EDWARDS(ed376, 7)
MODARITH_AMD_DECLARE_EDWARDS(ed500, 9)
/* NIST P-256, P-384, P-521, secp256k1 (a = 0, CONSTANT_B = 7, over pseudo.py's SECP256K1 field as curve.py builds it at
 * 64 bits) and NUMS256W (CONSTANT_B, CONSTANT_X) in short-Weierstrass form (weierstrass.c: complete add/dbl 68-281, setxy
 * 366-410, mul 494-543, mul2 545-569; constants curve.py:147-198) -- the same curve.h API and layouts; ecn_<c>_set
 * needs x (y optional), ecn_<c>_cof is a no-op.  Byte records whose length is a multiple of 8 must be 8-byte aligned; the 66-byte
 * P-521 records may sit anywhere. */
MODARITH_AMD_DECLARE_EDWARDS(nist256, 5)
MODARITH_AMD_DECLARE_EDWARDS(nist384, 7)
MODARITH_AMD_DECLARE_EDWARDS(nist521, 9)
MODARITH_AMD_DECLARE_EDWARDS(secp256k1, 5)
MODARITH_AMD_DECLARE_EDWARDS(nums256w, 5)

/* ---- Fused scalar multiplication + affine export: ecnXXXmul followed by ecnXXXget, the reference's own call
 * pattern (ed448.c:182-184: `ecnXXXmul(e,&P); ecnXXXget(&P,x,y)`), in one kernel.  x, y (device, big-endian Nbytes
 * records, either may be NULL) receive the affine coordinates of e*P, sign (device int[n] or NULL) the sign ecnXXXget
 * returns (of y when y is NULL, of x when x is NULL, else 0).  P is NOT modified (the two-call form leaves e*P in it).
 * Only canonical bytes leave the kernel, so it runs on 32-bit-limb internals with extended-coordinate formulas that are
 * complete on the curve (csrc/ed26.h): the same bytes as ecn_<c>_mul_batch + ecn_<c>_get_batch for every input point
 * on the curve whose coordinate limbs keep the limb budget -- every limb below 2^(Radix+2), which every point the library
 * (or field.c-style code) produced does; the fused kernels re-pack the 64-bit limbs into 32-bit ones and silently drop what
 * lies above (checkable beforehand with modlimbs_<P>_batch on the three coordinates; the plain ecn_<c>_mul_batch instead
 * reproduces the reference's 64-bit wrap-around for such fabricated limbs -- except for ed25519 and ed448, whose mul / mul2 /
 * mul2_exact keep the field elements in resident half-limb form since round 4 (csrc/fh51.h, fh56.h): their results are the
 * reference's limbs for every input inside the limb budget, i.e. every limb below 2^(Radix+2), and in fact for limbs up to
 * 2^58 (ed25519: the half-limb cut is exact up to there; beyond it the upper half is truncated to 32 bits) / for the same
 * INTEGER with the excess above 56 bits moved up (ed448: fh56.h from_limbs), not the reference's wrap-around).  This precondition holds for every fused entry
 * point below (mul_get, mul2_get, mulgen2_get).  Constant time like ecnXXXmul: ed448, nist256, secp256k1 by fixed windows with
 * scanned tables; ed25519 (round 5, csrc/ed26l.h) by a Montgomery ladder on the birationally equivalent curve with the second
 * coordinate recovered at the end and both inversions shared by up to 32 records -- no table at all.  workspace: a device buffer of
 * ecn_<c>_mul_get_workspace_bytes(n) bytes (ed448: the per-lane window tables, 672 bytes per resident lane, at most 88 MB;
 * ed25519: 140 bytes per record for at most 2^20 records at a time, i.e. at most 147 MB whatever n).  ed25519 only: with
 * workspace NULL (or too small) the call takes stream-ordered scratch from the library's own pool instead -- callers written
 * against rounds 2-4, where this function returned 0, keep working -- and fails with hipErrorInvalidValue only when that is
 * impossible (a stream under graph capture). */
size_t ecn_ed25519_mul_get_workspace_bytes(size_t n);
int ecn_ed25519_mul_get_batch(const char *e, const ma_spint *P, char *x, char *y, int *sign, size_t n, size_t ld,
                              void *workspace, size_t workspace_bytes, void *stream);
/* R = e*P + f*Q and its affine export in one kernel: ecnXXXmul2 followed by ecnXXXget, the verification pattern
 * (ed448.c:305, nist256.c:251-254); same conventions as mul_get, P and Q are not modified.  ed25519 / ed448 (round 5): a Straus
 * walk over signed 4-bit windows whose table entries are read BY INDEX (variable time like the reference's own mul2: public
 * inputs); the workspace (any address: the size reported includes the slack for the 128-byte alignment of the table lines) holds the table slabs of the resident grid (302 / 604 MB) and 140 / 236 bytes per
 * record for at most 2^20 records; NULL takes the library's scratch pool. */
size_t ecn_ed25519_mul2_get_workspace_bytes(size_t n);
int ecn_ed25519_mul2_get_batch(const char *e, const ma_spint *P, const char *f, const ma_spint *Q, char *x, char *y, int *sign,
                               size_t n, size_t ld, void *workspace, size_t workspace_bytes, void *stream);
size_t ecn_ed448_mul2_get_workspace_bytes(size_t n);
int ecn_ed448_mul2_get_batch(const char *e, const ma_spint *P, const char *f, const ma_spint *Q, char *x, char *y, int *sign,
                             size_t n, size_t ld, void *workspace, size_t workspace_bytes, void *stream);
size_t ecn_ed448_mul_get_workspace_bytes(size_t n);
int ecn_ed448_mul_get_batch(const char *e, const ma_spint *P, char *x, char *y, int *sign, size_t n, size_t ld,
                            void *workspace, size_t workspace_bytes, void *stream);
/* NIST P-256 (csrc/wn26.h, wj26.h): the ECDSA patterns nist256.c:155-161, 219-222 (ecnXXXmul + ecnXXXget) and nist256.c:251-256
 * (ecnXXXmul2 + ecnXXXget).  Ten signed 26-bit limbs, Montgomery form with R' = 2^286.  Round 5: the scalar is reduced mod the
 * group order and k P runs in Jacobian coordinates (doubling 3M + 5S against the 8M + 3S + 2 m_b of the complete formulas); where
 * the Jacobian addition could fail is decided by the scalar alone on a curve of prime order -- lane flags for the accumulator or the
 * digit at infinity, the LAST addition the complete one of weierstrass.c:68-175 -- so the affine bytes are the reference's for
 * every scalar and every point of the curve, the point at infinity included.  mul2_get (two caller points: its accumulator can meet
 * a table point anywhere) runs Jacobian mixed additions too, tests every one of them for that case and redoes it with the complete
 * formula: same bytes for every input, but a wave's instruction sequence then depends on the inputs -- public ones in a
 * verification; the reference's own mul2 (a joint sparse form) branches on them as well.  Workspace (ecn_<c>_*_get_workspace_bytes(n)): mul_get / mulgen2_get
 * keep the eight multiples of every record's point, 1 284 bytes per record for at most 2^19 records (673 MB): a first kernel computes
 * them, a second brings them to Z = 1 under an inversion shared by 32 entries, the window loop then runs mixed additions
 * (csrc/wn_affine.h); mul2_get the same with two tables of eight (2 564 bytes per record, 1.34 GB for a chunk).  Behind either: 160 bytes per record for at most 2^20 records -- (X : Y : Z) of the results, whose inversion is
 * shared by up to 32 records (csrc/wn_export.h).  A result at infinity leaves as x = 0, y = 1, the bytes
 * ecnXXXget produces for it (weierstrass.c:299-310).  Points off the curve mean nothing on either side and may differ. */
size_t ecn_nist256_mul_get_workspace_bytes(size_t n);
int ecn_nist256_mul_get_batch(const char *e, const ma_spint *P, char *x, char *y, int *sign, size_t n, size_t ld,
                              void *workspace, size_t workspace_bytes, void *stream);
size_t ecn_nist256_mul2_get_workspace_bytes(size_t n);
int ecn_nist256_mul2_get_batch(const char *e, const ma_spint *P, const char *f, const ma_spint *Q, char *x, char *y, int *sign,
                               size_t n, size_t ld, void *workspace, size_t workspace_bytes, void *stream);
/* secp256k1 (curve.py:190-198; csrc/wn26.h on csrc/fk26.h): the same patterns on the a = 0 complete formulas
 * (weierstrass.c:120-157, 189-226), ten signed 26-bit limbs with the pseudo-Mersenne fold 2^260 = 2^36 + 0x3d10.  Round 5
 * (csrc/glv26.h): every scalar is reduced mod the group order and split by the curve's endomorphism, k = k1 + k2 lambda with
 * |k1|, |k2| < 2^128, so that k P = k1 P + k2 (beta x, y) takes 128 doublings on the one table of P; all additions stay the
 * complete ones.  Same bytes as the two-call form for every scalar and every point of the curve.  Projective window tables per resident lane: 960 bytes (mul_get,
 * mulgen2_get: at most 189 MB), 1 920 (mul2_get: 377 MB), and the 160 bytes per record of the shared export as for P-256. */
size_t ecn_secp256k1_mul_get_workspace_bytes(size_t n);
int ecn_secp256k1_mul_get_batch(const char *e, const ma_spint *P, char *x, char *y, int *sign, size_t n, size_t ld,
                                void *workspace, size_t workspace_bytes, void *stream);
size_t ecn_secp256k1_mul2_get_workspace_bytes(size_t n);
int ecn_secp256k1_mul2_get_batch(const char *e, const ma_spint *P, const char *f, const ma_spint *Q, char *x, char *y, int *sign,
                                 size_t n, size_t ld, void *workspace, size_t workspace_bytes, void *stream);

/* Generator multiplication + affine export: ecnXXXgen, ecnXXXmul, ecnXXXget in one kernel -- the opening of
 * NIST256_KEY_PAIR and NIST256_SIGN (nist256.c:150-161, 214-222).  x, y, sign as for mul_get; there is no point
 * argument and no workspace: the multiples m * 32^i * G, m = 1..16, live in a 66 560-byte constant table
 * (generated/comb_<C>.h), every window reads all of its entries (constant-time) and adds one -- secp256k1 with the complete mixed
 * addition, P-256 (round 5) with the Jacobian one after reducing the scalar mod the group order (csrc/wj26.h: on the fixed-base
 * table the accumulator can meet the table point for no scalar below the order); one scalar per lane, the inversion shared by up
 * to 32 records through stream-ordered scratch of the library's pool (160 bytes per record for at most 2^20 records; on a stream
 * under capture, where that pool is not available: four scalars per lane sharing one inversion, as in rounds 2-4).  Same
 * bytes as ecn_<c>_gen_batch + ecn_<c>_mul_batch + ecn_<c>_get_batch for every 32-byte scalar. */
int ecn_nist256_mulgen_get_batch(const char *e, char *x, char *y, int *sign, size_t n, void *stream);
int ecn_secp256k1_mulgen_get_batch(const char *e, char *x, char *y, int *sign, size_t n, void *stream);
/* ED448_KEY_PAIR / ED448_SIGN open the same way (ed448.c:167-184, 196-199); ED25519: 65 4-bit windows, cached (y+x, y-x, 2dxy);
 * ED448: 113 windows, cached (x, y, 39081xy), 173 568-byte table */
int ecn_ed25519_mulgen_get_batch(const char *e, char *x, char *y, int *sign, size_t n, void *stream);
int ecn_ed448_mulgen_get_batch(const char *e, char *x, char *y, int *sign, size_t n, void *stream);

/* e*G + f*Q and its affine export: ecnXXXgen, ecnXXXmul2(e, &G, f, &Q, &R), ecnXXXget -- signature verification, where the
 * first point of the reference's double multiplication is always the generator (nist256.c:251-256, ed448.c:305).  f*Q as in
 * mul_get (workspace of ecn_<c>_mulgen2_get_workspace_bytes(n) bytes, rules as for mul_get), e*G through
 * the fixed-base table without doublings of its own; Q is not modified; same bytes as the three calls. */
size_t ecn_nist256_mulgen2_get_workspace_bytes(size_t n);
int ecn_nist256_mulgen2_get_batch(const char *e, const char *f, const ma_spint *Q, char *x, char *y, int *sign, size_t n, size_t ld,
                                  void *workspace, size_t workspace_bytes, void *stream);
size_t ecn_secp256k1_mulgen2_get_workspace_bytes(size_t n);
int ecn_secp256k1_mulgen2_get_batch(const char *e, const char *f, const ma_spint *Q, char *x, char *y, int *sign, size_t n, size_t ld,
                                    void *workspace, size_t workspace_bytes, void *stream);
size_t ecn_ed448_mulgen2_get_workspace_bytes(size_t n);   /* ED448_VERIFY, ed448.c:290-310 */
int ecn_ed448_mulgen2_get_batch(const char *e, const char *f, const ma_spint *Q, char *x, char *y, int *sign, size_t n, size_t ld,
                                void *workspace, size_t workspace_bytes, void *stream);
size_t ecn_ed25519_mulgen2_get_workspace_bytes(size_t n);
int ecn_ed25519_mulgen2_get_batch(const char *e, const char *f, const ma_spint *Q, char *x, char *y, int *sign, size_t n, size_t ld,
                                  void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MODARITH_AMD_H */
