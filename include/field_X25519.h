/* include/field_X25519.h -- EMITTED by modarith_amd/emit.py field_shim_text(); do not edit.
 *
 * Put  #include "field_X25519.h"  where the reference's templates say "paste field.c here" (rfc7748.c:24-28,
 * edwards.c:19-23 @field@, weierstrass.c:16-20, edge.c:5-9; automated there by curve.py:335-351) and link
 * libmodarith_amd.so: the template's calls modmul(a, b, c) ... then run
 * on the GPU one element at a time (host pointers, the reference's signatures and aliasing rules; a bring-up path --
 * throughput comes from the <fn>_X25519_batch entry points of modarith_amd.h).
 * prime X25519 = 0x7fffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffed, pseudo.py form
 */
#ifndef MODARITH_AMD_FIELD_X25519_H
#define MODARITH_AMD_FIELD_X25519_H
#include <stdio.h>
#include <stdint.h>
#include "modarith_amd.h"
/* (modarith_amd.h declares the X25519 entry points) */

#define sspint int64_t
#define spint uint64_t
#define dpint __uint128_t
#define sdpint __int128_t
#define Wordlength 64
#define Nlimbs 5
#define Radix 51
#define Nbits 255
#define Nbytes 32

#define MERSENNE
#define MULBYINT
#define X25519

#define prop prop_X25519_ct
#define flatten flatten_X25519_ct
#define modfsb modfsb_X25519_ct
#define modadd modadd_X25519_ct
#define modsub modsub_X25519_ct
#define modneg modneg_X25519_ct
#define modmli modmli_X25519_ct
#define modmul modmul_X25519_ct
#define modsqr modsqr_X25519_ct
#define modcpy modcpy_X25519_ct
#define modnsqr modnsqr_X25519_ct
#define modpro modpro_X25519_ct
#define modinv modinv_X25519_ct
#define nres nres_X25519_ct
#define redc redc_X25519_ct
#define modis1 modis1_X25519_ct
#define modis0 modis0_X25519_ct
#define modzer modzer_X25519_ct
#define modone modone_X25519_ct
#define modint modint_X25519_ct
#define modqr modqr_X25519_ct
#define modcmv modcmv_X25519_ct
#define modcsw modcsw_X25519_ct
#define modsqrt modsqrt_X25519_ct
#define modshl modshl_X25519_ct
#define modshr modshr_X25519_ct
#define modhaf modhaf_X25519_ct
#define mod2r mod2r_X25519_ct
#define modexp modexp_X25519_ct
#define modimp modimp_X25519_ct
#define modsign modsign_X25519_ct
#define modcmp modcmp_X25519_ct

#endif
