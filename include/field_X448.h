/* include/field_X448.h -- EMITTED by modarith_amd/emit.py field_shim_text(); do not edit.
 *
 * Put  #include "field_X448.h"  where the reference's templates say "paste field.c here" (rfc7748.c:24-28,
 * edwards.c:19-23 @field@, weierstrass.c:16-20, edge.c:5-9; automated there by curve.py:335-351) and link
 * libmodarith_amd.so: the template's calls modmul(a, b, c) ... then run
 * on the GPU one element at a time (host pointers, the reference's signatures and aliasing rules; a bring-up path --
 * throughput comes from the <fn>_X448_batch entry points of modarith_amd.h).
 * prime X448 = 0xfffffffffffffffffffffffffffffffffffffffffffffffffffffffeffffffffffffffffffffffffffffffffffffffffffffffffffffffff, monty.py form
 */
#ifndef MODARITH_AMD_FIELD_X448_H
#define MODARITH_AMD_FIELD_X448_H
#include <stdio.h>
#include <stdint.h>
#include "modarith_amd.h"
/* (modarith_amd.h declares the X448 entry points) */

#define sspint int64_t
#define spint uint64_t
#define dpint __uint128_t
#define sdpint __int128_t
#define Wordlength 64
#define Nlimbs 8
#define Radix 56
#define Nbits 448
#define Nbytes 56

#define MONTGOMERY
#define X448
#define MULBYINT

#define prop prop_X448_ct
#define flatten flatten_X448_ct
#define modfsb modfsb_X448_ct
#define modadd modadd_X448_ct
#define modsub modsub_X448_ct
#define modneg modneg_X448_ct
#define modmli modmli_X448_ct
#define modmul modmul_X448_ct
#define modsqr modsqr_X448_ct
#define modcpy modcpy_X448_ct
#define modnsqr modnsqr_X448_ct
#define modpro modpro_X448_ct
#define modinv modinv_X448_ct
#define nres nres_X448_ct
#define redc redc_X448_ct
#define modis1 modis1_X448_ct
#define modis0 modis0_X448_ct
#define modzer modzer_X448_ct
#define modone modone_X448_ct
#define modint modint_X448_ct
#define modqr modqr_X448_ct
#define modcmv modcmv_X448_ct
#define modcsw modcsw_X448_ct
#define modsqrt modsqrt_X448_ct
#define modshl modshl_X448_ct
#define modshr modshr_X448_ct
#define modhaf modhaf_X448_ct
#define mod2r mod2r_X448_ct
#define modexp modexp_X448_ct
#define modimp modimp_X448_ct
#define modsign modsign_X448_ct
#define modcmp modcmp_X448_ct

#endif
