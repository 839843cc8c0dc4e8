/* include/field_NIST256.h -- EMITTED by modarith_amd/emit.py field_shim_text(); do not edit.
 *
 * Put  #include "field_NIST256.h"  where the reference's templates say "paste field.c here" (rfc7748.c:24-28,
 * edwards.c:19-23 @field@, weierstrass.c:16-20, edge.c:5-9; automated there by curve.py:335-351) and link
 * libmodarith_amd.so: the template's calls modmul(a, b, c) ... then run
 * on the GPU one element at a time (host pointers, the reference's signatures and aliasing rules; a bring-up path --
 * throughput comes from the <fn>_NIST256_batch entry points of modarith_amd.h).
 * prime NIST256 = 0xffffffff00000001000000000000000000000000ffffffffffffffffffffffff, monty.py form
 */
#ifndef MODARITH_AMD_FIELD_NIST256_H
#define MODARITH_AMD_FIELD_NIST256_H
#include <stdio.h>
#include <stdint.h>
#include "modarith_amd.h"
/* (modarith_amd.h declares the NIST256 entry points) */

#define sspint int64_t
#define spint uint64_t
#define dpint __uint128_t
#define sdpint __int128_t
#define Wordlength 64
#define Nlimbs 5
#define Radix 52
#define Nbits 256
#define Nbytes 32

#define MONTGOMERY
#define NIST256

#define prop prop_NIST256_ct
#define flatten flatten_NIST256_ct
#define modfsb modfsb_NIST256_ct
#define modadd modadd_NIST256_ct
#define modsub modsub_NIST256_ct
#define modneg modneg_NIST256_ct
#define modmli modmli_NIST256_ct
#define modmul modmul_NIST256_ct
#define modsqr modsqr_NIST256_ct
#define modcpy modcpy_NIST256_ct
#define modnsqr modnsqr_NIST256_ct
#define modpro modpro_NIST256_ct
#define modinv modinv_NIST256_ct
#define nres nres_NIST256_ct
#define redc redc_NIST256_ct
#define modis1 modis1_NIST256_ct
#define modis0 modis0_NIST256_ct
#define modzer modzer_NIST256_ct
#define modone modone_NIST256_ct
#define modint modint_NIST256_ct
#define modqr modqr_NIST256_ct
#define modcmv modcmv_NIST256_ct
#define modcsw modcsw_NIST256_ct
#define modsqrt modsqrt_NIST256_ct
#define modshl modshl_NIST256_ct
#define modshr modshr_NIST256_ct
#define modhaf modhaf_NIST256_ct
#define mod2r mod2r_NIST256_ct
#define modexp modexp_NIST256_ct
#define modimp modimp_NIST256_ct
#define modsign modsign_NIST256_ct
#define modcmp modcmp_NIST256_ct

#endif
