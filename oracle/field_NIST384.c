/* oracle/field_NIST384.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * The field.c function set for NIST P-384 (`python3 monty.py 64 NIST384`: 7 limbs of 56 bits, R = 2^392, full
 * Montgomery digits) so that the curve-layer restatement (weierstrass_body.inc) can be instantiated on it.  The six
 * prime-specific functions are the run-time generic oracle (field_generic.c) bound to the parameter block that
 * tests fill from the constants captured from the reference (tests/golden/field_NIST384.json "params", against
 * whose vectors the generic oracle is pinned by tests/test_generic_oracle.py); the rest is field_common.inc.
 * oracle_bind_NIST384() must be called once before anything else here.
 */
#include "oracle_types.h"
#include "field_generic.h"
#define PRIME NIST384
#define ORACLE_MONTGOMERY
#define NL 7
#define RADIX 56
#define NBITS 384
#define NBYTES 48
#define PM1D2 1

static gparams bound;
int oracle_bind_NIST384(const gparams *p) {
    if (p->n != NL || p->radix != RADIX || p->nbits != NBITS || p->nbytes != NBYTES || p->pm1d2 != PM1D2) return 1;
    bound = *p;
    return 0;
}
#define PP_CNT (bound.pp_cnt)
#define pp_idx (bound.pp_idx)
#define pp_sgn (bound.pp_sgn)
#define pp_val (bound.pp_val)
#define roi (bound.roi)

void modmul_NIST384(const spint *a, const spint *b, spint *c) { gen_modmul(&bound, a, b, c); }
void modsqr_NIST384(const spint *a, spint *c) { gen_modsqr(&bound, a, c); }
void modmli_NIST384(const spint *a, int b, spint *c) { gen_modmli(&bound, a, b, c); }
void nres_NIST384(const spint *m, spint *n) { gen_nres(&bound, m, n); }
void redc_NIST384(const spint *n, spint *m) { gen_redc(&bound, n, m); }
void modpro_NIST384(const spint *w, spint *z) { gen_modpro(&bound, w, z); }

#include "field_common.inc"
