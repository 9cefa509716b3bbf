/* CPU oracle internals.  TEST INFRASTRUCTURE ONLY (see orc.h). */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H
#include "constants.h"
#include "orc.h"

/* the field types ARE the ABI types (no type punning) */
typedef orc_fr fr_t;
typedef orc_fq fq_t;
#define FE_EXTERNAL

/* Fr: 4 limbs */
#define FE fr_t
#define FN(x) fr_##x
#define NL 4
#define FMOD ORC_FR_MODULUS
#define FINV ORC_FR_INV
#define FONE ORC_FR_ONE
#define FR2 ORC_FR_R2
#include "mont_tmpl.h"
#undef FE
#undef FN
#undef NL
#undef FMOD
#undef FINV
#undef FONE
#undef FR2

/* Fq: 6 limbs */
#define FE fq_t
#define FN(x) fq_##x
#define NL 6
#define FMOD ORC_FQ_MODULUS
#define FINV ORC_FQ_INV
#define FONE ORC_FQ_ONE
#define FR2 ORC_FQ_R2
#include "mont_tmpl.h"
#undef FE
#undef FN
#undef NL
#undef FMOD
#undef FINV
#undef FONE
#undef FR2

#define F(p) ((fq_t *)(p))
#define S(p) ((fr_t *)(p))

void fq2_add(orc_fq2 *r, const orc_fq2 *a, const orc_fq2 *b);
void fq2_sub(orc_fq2 *r, const orc_fq2 *a, const orc_fq2 *b);
void fq2_neg(orc_fq2 *r, const orc_fq2 *a);
void fq2_dbl(orc_fq2 *r, const orc_fq2 *a);
void fq2_mul(orc_fq2 *r, const orc_fq2 *a, const orc_fq2 *b);
void fq2_sqr(orc_fq2 *r, const orc_fq2 *a);
void fq2_inv(orc_fq2 *r, const orc_fq2 *a);
void fq2_mul_fq(orc_fq2 *r, const orc_fq2 *a, const fq_t *s);
void fq2_mul_nonres(orc_fq2 *r, const orc_fq2 *a);
int fq2_is_zero(const orc_fq2 *a);
int fq2_eq(const orc_fq2 *a, const orc_fq2 *b);
void fq12_sqr(orc_fq12 *r, const orc_fq12 *a);
void fq12_conj(orc_fq12 *r, const orc_fq12 *a);

#endif
