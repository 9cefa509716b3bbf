/* CPU oracle: fields, tower, G1/G2, ate pairing.  TEST INFRASTRUCTURE ONLY (see orc.h).
 *
 * Restates (published algorithms, not copied code; the snarkVM sources are not available):
 *   snarkvm_fields Fp256/Fp384 Montgomery arithmetic            
 *   snarkvm_curves::bls12_377 G1Affine/G1Projective, G2           
 *   Bls12_377::{miller_loop, final_exponentiation}                
 * Tower: Fq2 = Fq[u]/(u^2+5), Fq6 = Fq2[v]/(v^3-u), Fq12 = Fq6[w]/(w^2-v)  (SURVEY.md Appendix A).
 */
#include "orc_internal.h"
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ exported field wrappers */
void orc_fr_mul(orc_fr *r, const orc_fr *a, const orc_fr *b) { fr_mul((fr_t *)r, (const fr_t *)a, (const fr_t *)b); }
void orc_fr_add(orc_fr *r, const orc_fr *a, const orc_fr *b) { fr_add((fr_t *)r, (const fr_t *)a, (const fr_t *)b); }
void orc_fr_sub(orc_fr *r, const orc_fr *a, const orc_fr *b) { fr_sub((fr_t *)r, (const fr_t *)a, (const fr_t *)b); }
void orc_fr_inv(orc_fr *r, const orc_fr *a) { fr_inv((fr_t *)r, (const fr_t *)a); }
void orc_fr_from_canonical(orc_fr *r, const orc_bigint256 *a) { fr_from_canonical((fr_t *)r, a->l); }
void orc_fr_to_canonical(orc_bigint256 *r, const orc_fr *a) { fr_to_canonical(r->l, (const fr_t *)a); }
void orc_fq_mul(orc_fq *r, const orc_fq *a, const orc_fq *b) { fq_mul((fq_t *)r, (const fq_t *)a, (const fq_t *)b); }
void orc_fq_add(orc_fq *r, const orc_fq *a, const orc_fq *b) { fq_add((fq_t *)r, (const fq_t *)a, (const fq_t *)b); }
void orc_fq_sub(orc_fq *r, const orc_fq *a, const orc_fq *b) { fq_sub((fq_t *)r, (const fq_t *)a, (const fq_t *)b); }
void orc_fq_inv(orc_fq *r, const orc_fq *a) { fq_inv((fq_t *)r, (const fq_t *)a); }
void orc_fq_from_canonical(orc_fq *r, const uint64_t *l) { fq_from_canonical((fq_t *)r, l); }
void orc_fq_to_canonical(uint64_t *l, const orc_fq *a) { fq_to_canonical(l, (const fq_t *)a); }

/* ------------------------------------------------------------------ Fq2 */
static inline void fq_mul5(fq_t *r, const fq_t *a) {
    fq_t t;
    fq_dbl(&t, a);
    fq_dbl(&t, &t);
    fq_add(r, &t, a);
}
void fq2_add(orc_fq2 *r, const orc_fq2 *a, const orc_fq2 *b) {
    fq_add(F(&r->c0), F(&a->c0), F(&b->c0));
    fq_add(F(&r->c1), F(&a->c1), F(&b->c1));
}
void fq2_sub(orc_fq2 *r, const orc_fq2 *a, const orc_fq2 *b) {
    fq_sub(F(&r->c0), F(&a->c0), F(&b->c0));
    fq_sub(F(&r->c1), F(&a->c1), F(&b->c1));
}
void fq2_neg(orc_fq2 *r, const orc_fq2 *a) {
    fq_neg(F(&r->c0), F(&a->c0));
    fq_neg(F(&r->c1), F(&a->c1));
}
void fq2_dbl(orc_fq2 *r, const orc_fq2 *a) { fq2_add(r, a, a); }
int fq2_is_zero(const orc_fq2 *a) { return fq_is_zero(F(&a->c0)) && fq_is_zero(F(&a->c1)); }
int fq2_eq(const orc_fq2 *a, const orc_fq2 *b) { return fq_eq(F(&a->c0), F(&b->c0)) && fq_eq(F(&a->c1), F(&b->c1)); }
void fq2_mul(orc_fq2 *r, const orc_fq2 *a, const orc_fq2 *b) {
    fq_t t0, t1, s0, s1, m, t5;
    fq_mul(&t0, F(&a->c0), F(&b->c0));
    fq_mul(&t1, F(&a->c1), F(&b->c1));
    fq_add(&s0, F(&a->c0), F(&a->c1));
    fq_add(&s1, F(&b->c0), F(&b->c1));
    fq_mul(&m, &s0, &s1);
    fq_sub(&m, &m, &t0);
    fq_sub(F(&r->c1), &m, &t1);
    fq_mul5(&t5, &t1);
    fq_sub(F(&r->c0), &t0, &t5);
}
void fq2_sqr(orc_fq2 *r, const orc_fq2 *a) { fq2_mul(r, a, a); }
void fq2_mul_fq(orc_fq2 *r, const orc_fq2 *a, const fq_t *s) {
    fq_mul(F(&r->c0), F(&a->c0), s);
    fq_mul(F(&r->c1), F(&a->c1), s);
}
/* times u: (c0 + c1 u) u = -5 c1 + c0 u */
void fq2_mul_nonres(orc_fq2 *r, const orc_fq2 *a) {
    fq_t t, c0 = *F(&a->c0);
    fq_mul5(&t, F(&a->c1));
    fq_neg(F(&r->c0), &t);
    *F(&r->c1) = c0;
}
void fq2_inv(orc_fq2 *r, const orc_fq2 *a) {
    fq_t n, t, t5;
    fq_sqr(&n, F(&a->c0));
    fq_sqr(&t, F(&a->c1));
    fq_mul5(&t5, &t);
    fq_add(&n, &n, &t5);
    fq_inv(&n, &n);
    fq_mul(F(&r->c0), F(&a->c0), &n);
    fq_mul(&t, F(&a->c1), &n);
    fq_neg(F(&r->c1), &t);
}

/* ------------------------------------------------------------------ Fq6 */
static void fq6_add(orc_fq6 *r, const orc_fq6 *a, const orc_fq6 *b) {
    fq2_add(&r->c0, &a->c0, &b->c0);
    fq2_add(&r->c1, &a->c1, &b->c1);
    fq2_add(&r->c2, &a->c2, &b->c2);
}
static void fq6_sub(orc_fq6 *r, const orc_fq6 *a, const orc_fq6 *b) {
    fq2_sub(&r->c0, &a->c0, &b->c0);
    fq2_sub(&r->c1, &a->c1, &b->c1);
    fq2_sub(&r->c2, &a->c2, &b->c2);
}
static void fq6_neg(orc_fq6 *r, const orc_fq6 *a) {
    fq2_neg(&r->c0, &a->c0);
    fq2_neg(&r->c1, &a->c1);
    fq2_neg(&r->c2, &a->c2);
}
/* times v: (c0, c1, c2) -> (u c2, c0, c1) */
static void fq6_mul_nonres(orc_fq6 *r, const orc_fq6 *a) {
    orc_fq2 t;
    fq2_mul_nonres(&t, &a->c2);
    orc_fq2 c0 = a->c0, c1 = a->c1;
    r->c0 = t;
    r->c1 = c0;
    r->c2 = c1;
}
static void fq6_mul(orc_fq6 *r, const orc_fq6 *a, const orc_fq6 *b) {
    orc_fq2 v0, v1, v2, s, t, x, c0, c1, c2;
    fq2_mul(&v0, &a->c0, &b->c0);
    fq2_mul(&v1, &a->c1, &b->c1);
    fq2_mul(&v2, &a->c2, &b->c2);
    fq2_add(&s, &a->c1, &a->c2);
    fq2_add(&t, &b->c1, &b->c2);
    fq2_mul(&x, &s, &t);
    fq2_sub(&x, &x, &v1);
    fq2_sub(&x, &x, &v2);
    fq2_mul_nonres(&x, &x);
    fq2_add(&c0, &x, &v0);
    fq2_add(&s, &a->c0, &a->c1);
    fq2_add(&t, &b->c0, &b->c1);
    fq2_mul(&x, &s, &t);
    fq2_sub(&x, &x, &v0);
    fq2_sub(&x, &x, &v1);
    fq2_mul_nonres(&s, &v2);
    fq2_add(&c1, &x, &s);
    fq2_add(&s, &a->c0, &a->c2);
    fq2_add(&t, &b->c0, &b->c2);
    fq2_mul(&x, &s, &t);
    fq2_sub(&x, &x, &v0);
    fq2_sub(&x, &x, &v2);
    fq2_add(&c2, &x, &v1);
    r->c0 = c0;
    r->c1 = c1;
    r->c2 = c2;
}
/* a * (b0 + b1 v) */
static void fq6_mul_by_01(orc_fq6 *r, const orc_fq6 *a, const orc_fq2 *b0, const orc_fq2 *b1) {
    orc_fq2 t0, t1, c0, c1, c2;
    fq2_mul(&t0, &a->c0, b0);
    fq2_mul(&t1, &a->c2, b1);
    fq2_mul_nonres(&t1, &t1);
    fq2_add(&c0, &t0, &t1);
    fq2_mul(&t0, &a->c0, b1);
    fq2_mul(&t1, &a->c1, b0);
    fq2_add(&c1, &t0, &t1);
    fq2_mul(&t0, &a->c1, b1);
    fq2_mul(&t1, &a->c2, b0);
    fq2_add(&c2, &t0, &t1);
    r->c0 = c0;
    r->c1 = c1;
    r->c2 = c2;
}
static void fq6_inv(orc_fq6 *r, const orc_fq6 *a) {
    orc_fq2 t0, t1, t2, x, d;
    fq2_sqr(&t0, &a->c0);
    fq2_mul(&x, &a->c1, &a->c2);
    fq2_mul_nonres(&x, &x);
    fq2_sub(&t0, &t0, &x);
    fq2_sqr(&t1, &a->c2);
    fq2_mul_nonres(&t1, &t1);
    fq2_mul(&x, &a->c0, &a->c1);
    fq2_sub(&t1, &t1, &x);
    fq2_sqr(&t2, &a->c1);
    fq2_mul(&x, &a->c0, &a->c2);
    fq2_sub(&t2, &t2, &x);
    fq2_mul(&d, &a->c2, &t1);
    fq2_mul(&x, &a->c1, &t2);
    fq2_add(&d, &d, &x);
    fq2_mul_nonres(&d, &d);
    fq2_mul(&x, &a->c0, &t0);
    fq2_add(&d, &d, &x);
    fq2_inv(&d, &d);
    fq2_mul(&r->c0, &t0, &d);
    fq2_mul(&r->c1, &t1, &d);
    fq2_mul(&r->c2, &t2, &d);
}

/* ------------------------------------------------------------------ Fq12 */
void orc_fq12_one(orc_fq12 *r) {
    memset(r, 0, sizeof *r);
    fq_one(F(&r->c0.c0.c0));
}
int orc_fq12_is_one(const orc_fq12 *a) {
    orc_fq12 o;
    orc_fq12_one(&o);
    return memcmp(a, &o, sizeof o) == 0;
}
void orc_fq12_mul(orc_fq12 *r, const orc_fq12 *a, const orc_fq12 *b) {
    orc_fq6 v0, v1, s, t, x;
    fq6_mul(&v0, &a->c0, &b->c0);
    fq6_mul(&v1, &a->c1, &b->c1);
    fq6_add(&s, &a->c0, &a->c1);
    fq6_add(&t, &b->c0, &b->c1);
    fq6_mul(&x, &s, &t);
    fq6_sub(&x, &x, &v0);
    fq6_sub(&r->c1, &x, &v1);
    fq6_mul_nonres(&v1, &v1);
    fq6_add(&r->c0, &v0, &v1);
}
void fq12_sqr(orc_fq12 *r, const orc_fq12 *a) {
    /* complex squaring: c0 = (a0+a1)(a0+v a1) - a0a1 - v a0a1 ; c1 = 2 a0a1 */
    orc_fq6 ab, s, t, vt;
    fq6_mul(&ab, &a->c0, &a->c1);
    fq6_add(&s, &a->c0, &a->c1);
    fq6_mul_nonres(&vt, &a->c1);
    fq6_add(&t, &a->c0, &vt);
    fq6_mul(&s, &s, &t);
    fq6_sub(&s, &s, &ab);
    fq6_mul_nonres(&vt, &ab);
    fq6_sub(&r->c0, &s, &vt);
    fq6_add(&r->c1, &ab, &ab);
}
void fq12_conj(orc_fq12 *r, const orc_fq12 *a) {
    r->c0 = a->c0;
    fq6_neg(&r->c1, &a->c1);
}
void orc_fq12_inv(orc_fq12 *r, const orc_fq12 *a) {
    orc_fq6 t0, t1;
    fq6_mul(&t0, &a->c0, &a->c0);
    fq6_mul(&t1, &a->c1, &a->c1);
    fq6_mul_nonres(&t1, &t1);
    fq6_sub(&t0, &t0, &t1);
    fq6_inv(&t0, &t0);
    fq6_mul(&r->c0, &a->c0, &t0);
    fq6_mul(&t1, &a->c1, &t0);
    fq6_neg(&r->c1, &t1);
}
void orc_fq12_frobenius(orc_fq12 *r, const orc_fq12 *a, int power) {
    /* coefficient of w^k (k = i + 2j + 6b for c_i.c_j.c_b) is in Fq and is multiplied by gamma_e^k;
     * for odd e the u-part additionally conjugates, which gamma^6 = -1 already encodes. */
    const orc_fq6 *ai[2] = {&a->c0, &a->c1};
    orc_fq6 *ri[2] = {&r->c0, &r->c1};
    for (int i = 0; i < 2; i++) {
        const orc_fq2 *aj[3] = {&ai[i]->c0, &ai[i]->c1, &ai[i]->c2};
        orc_fq2 *rj[3] = {&ri[i]->c0, &ri[i]->c1, &ri[i]->c2};
        for (int j = 0; j < 3; j++) {
            int k0 = i + 2 * j, k1 = k0 + 6;
            fq_mul(F(&rj[j]->c0), F(&aj[j]->c0), (const fq_t *)ORC_FROB[power - 1][k0]);
            fq_mul(F(&rj[j]->c1), F(&aj[j]->c1), (const fq_t *)ORC_FROB[power - 1][k1]);
        }
    }
}

/* f *= (yP + b w + c w^3), b,c in Fq2, yP in Fq:  in the tower l0 = (yP,0,0), l1 = (b,c,0) */
static void fq12_mul_by_line(orc_fq12 *f, const fq_t *yp, const orc_fq2 *b, const orc_fq2 *c) {
    orc_fq6 A, B, C, s;
    orc_fq2 b2;
    fq2_mul_fq(&A.c0, &f->c0.c0, yp);
    fq2_mul_fq(&A.c1, &f->c0.c1, yp);
    fq2_mul_fq(&A.c2, &f->c0.c2, yp);
    fq6_mul_by_01(&B, &f->c1, b, c);
    fq6_add(&s, &f->c0, &f->c1);
    b2 = *b;
    fq_add(F(&b2.c0), F(&b2.c0), yp);
    fq6_mul_by_01(&C, &s, &b2, c);
    fq6_sub(&C, &C, &A);
    fq6_sub(&f->c1, &C, &B);
    fq6_mul_nonres(&B, &B);
    fq6_add(&f->c0, &A, &B);
}

/* ------------------------------------------------------------------ G1 */
void orc_g1_generator(orc_g1_affine *g) {
    memcpy(&g->x, ORC_G1_GEN_X, 48);
    memcpy(&g->y, ORC_G1_GEN_Y, 48);
}
static int g1a_is_inf(const orc_g1_affine *p) { return fq_is_zero(F(&p->x)) && fq_is_zero(F(&p->y)); }
int orc_g1_is_on_curve(const orc_g1_affine *p) {
    if (g1a_is_inf(p)) return 1;
    fq_t l, r, one;
    fq_sqr(&l, F(&p->y));
    fq_sqr(&r, F(&p->x));
    fq_mul(&r, &r, F(&p->x));
    fq_one(&one);
    fq_add(&r, &r, &one);
    return fq_eq(&l, &r);
}
/* ---- compressed G1 points: snarkVM's CanonicalSerialize of a short-Weierstrass affine point, compressed [RECALL:
 * snarkvm-curves templates/short_weierstrass_jacobian + SWFlags, not in /root/reference]: 48 bytes, x little-endian and
 * below q; bit 7 of the last byte = y is the larger root (y > (q - 1) / 2), bit 6 = infinity (x = 0).  parity unpinned. */
static void fq_pow_limbs(fq_t *r, const fq_t *a, const uint64_t *e, int n_limbs) {
    fq_t acc;
    fq_one(&acc);
    for (int i = 64 * n_limbs - 1; i >= 0; i--) {
        fq_sqr(&acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) fq_mul(&acc, &acc, a);
    }
    *r = acc;
}
/* Tonelli-Shanks over the 2^46-th roots of unity; returns 0 when a is not a square */
int orc_fq_sqrt(orc_fq *out, const orc_fq *a_in) {
    const fq_t *a = (const fq_t *)a_in;
    fq_t w, x, b, c, one;
    fq_one(&one);
    if (fq_is_zero(a)) { *(fq_t *)out = *a; return 1; }
    fq_pow_limbs(&w, a, ORC_FQ_T_MINUS1_HALF, 6); /* a^((t-1)/2) */
    fq_mul(&x, a, &w);                            /* a^((t+1)/2) */
    fq_mul(&b, &x, &w);                           /* a^t */
    memcpy(&c, ORC_FQ_2SYLOW_GEN, sizeof c);
    int m = ORC_FQ_TWO_ADICITY;
    while (!fq_eq(&b, &one)) {
        int i = 0;
        fq_t b2 = b;
        while (!fq_eq(&b2, &one)) {
            fq_sqr(&b2, &b2);
            if (++i == m) return 0; /* b has order 2^m: a is not a square */
        }
        fq_t g = c;
        for (int k = 0; k < m - i - 1; k++) fq_sqr(&g, &g);
        fq_mul(&x, &x, &g);
        fq_sqr(&c, &g);
        fq_mul(&b, &b, &c);
        m = i;
    }
    *(fq_t *)out = x;
    return 1;
}
static int fq_is_larger_root(const fq_t *y) {
    uint64_t c[6];
    orc_fq_to_canonical(c, (const orc_fq *)y);
    for (int i = 5; i >= 0; i--) {
        if (c[i] > ORC_FQ_HALF[i]) return 1;
        if (c[i] < ORC_FQ_HALF[i]) return 0;
    }
    return 0;
}
void orc_g1_compress(uint8_t out[48], const orc_g1_affine *p) {
    memset(out, 0, 48);
    if (g1a_is_inf(p)) { out[47] = 0x40; return; }
    uint64_t c[6];
    orc_fq_to_canonical(c, &p->x);
    for (int i = 0; i < 48; i++) out[i] = (uint8_t)(c[i / 8] >> (8 * (i % 8)));
    if (fq_is_larger_root(F(&p->y))) out[47] |= 0x80;
}
/* r P for ANY curve point (the scalar is the group order itself, not reduced): is the result infinity? */
static int g1_in_subgroup(const orc_g1_affine *p) {
    orc_g1_jac acc;
    orc_g1_affine inf;
    memset(&inf, 0, sizeof inf);
    orc_g1_from_affine(&acc, &inf);
    for (int i = 252; i >= 0; i--) {
        orc_g1_double(&acc, &acc);
        if ((ORC_FR_MODULUS[i / 64] >> (i % 64)) & 1) orc_g1_add_mixed(&acc, &acc, p);
    }
    return fq_is_zero(F(&acc.z));
}
int orc_g1_decompress(orc_g1_affine *out, const uint8_t in[48], int check_subgroup) {
    memset(out, 0, sizeof *out);
    const uint8_t flags = in[47] & 0xC0;
    uint64_t c[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 48; i++) c[i / 8] |= (uint64_t)(i == 47 ? (in[i] & 0x3F) : in[i]) << (8 * (i % 8));
    int zero = 1;
    for (int i = 0; i < 6; i++) zero = zero && c[i] == 0;
    if (flags & 0x40) return zero && !(flags & 0x80);
    for (int i = 5; i >= 0; i--) { /* x < q */
        if (c[i] < ORC_FQ_MODULUS[i]) break;
        if (c[i] > ORC_FQ_MODULUS[i] || i == 0) return 0;
    }
    fq_t x, rhs, one, y;
    orc_fq_from_canonical((orc_fq *)&x, c);
    fq_one(&one);
    fq_sqr(&rhs, &x);
    fq_mul(&rhs, &rhs, &x);
    fq_add(&rhs, &rhs, &one);
    if (!orc_fq_sqrt((orc_fq *)&y, (const orc_fq *)&rhs)) return 0;
    fq_t chk;
    fq_sqr(&chk, &y);
    if (!fq_eq(&chk, &rhs)) return 0;
    if (fq_is_zero(&y) && (flags & 0x80)) return 0; /* y = 0 is its own negative: the sign flag selects nothing (non-canonical) */
    if (fq_is_larger_root(&y) != ((flags & 0x80) != 0)) fq_neg(&y, &y);
    *F(&out->x) = x;
    *F(&out->y) = y;
    if (check_subgroup && !g1_in_subgroup(out)) { memset(out, 0, sizeof *out); return 0; }
    return 1;
}
void orc_g1_neg_affine(orc_g1_affine *r, const orc_g1_affine *a) {
    r->x = a->x;
    fq_neg(F(&r->y), F(&a->y));
}
void orc_g1_from_affine(orc_g1_jac *r, const orc_g1_affine *a) {
    if (g1a_is_inf(a)) {
        memset(r, 0, sizeof *r);
        fq_one(F(&r->x));
        fq_one(F(&r->y));
        return;
    }
    r->x = a->x;
    r->y = a->y;
    fq_one(F(&r->z));
}
void orc_g1_double(orc_g1_jac *r, const orc_g1_jac *p) {
    if (fq_is_zero(F(&p->z))) { *r = *p; return; }
    fq_t A, B, C, D, E, Fv, t, x3, y3, z3;
    fq_sqr(&A, F(&p->x));
    fq_sqr(&B, F(&p->y));
    fq_sqr(&C, &B);
    fq_add(&t, F(&p->x), &B);
    fq_sqr(&t, &t);
    fq_sub(&t, &t, &A);
    fq_sub(&t, &t, &C);
    fq_dbl(&D, &t);
    fq_dbl(&E, &A);
    fq_add(&E, &E, &A);
    fq_sqr(&Fv, &E);
    fq_dbl(&t, &D);
    fq_sub(&x3, &Fv, &t);
    fq_sub(&t, &D, &x3);
    fq_mul(&y3, &E, &t);
    fq_dbl(&t, &C);
    fq_dbl(&t, &t);
    fq_dbl(&t, &t);
    fq_sub(&y3, &y3, &t);
    fq_mul(&z3, F(&p->y), F(&p->z));
    fq_dbl(&z3, &z3);
    *F(&r->x) = x3;
    *F(&r->y) = y3;
    *F(&r->z) = z3;
}
void orc_g1_add_mixed(orc_g1_jac *r, const orc_g1_jac *p, const orc_g1_affine *q) {
    if (g1a_is_inf(q)) { *r = *p; return; }
    if (fq_is_zero(F(&p->z))) { orc_g1_from_affine(r, q); return; }
    fq_t z1z1, u2, s2, h, hh, i, j, rr, v, t, x3, y3, z3;
    fq_sqr(&z1z1, F(&p->z));
    fq_mul(&u2, F(&q->x), &z1z1);
    fq_mul(&s2, F(&q->y), F(&p->z));
    fq_mul(&s2, &s2, &z1z1);
    fq_sub(&h, &u2, F(&p->x));
    fq_sub(&rr, &s2, F(&p->y));
    if (fq_is_zero(&h)) {
        if (fq_is_zero(&rr)) { orc_g1_double(r, p); return; }
        memset(r, 0, sizeof *r);
        fq_one(F(&r->x));
        fq_one(F(&r->y));
        return;
    }
    fq_sqr(&hh, &h);
    fq_dbl(&i, &hh);
    fq_dbl(&i, &i);
    fq_mul(&j, &h, &i);
    fq_dbl(&rr, &rr);
    fq_mul(&v, F(&p->x), &i);
    fq_sqr(&x3, &rr);
    fq_sub(&x3, &x3, &j);
    fq_sub(&x3, &x3, &v);
    fq_sub(&x3, &x3, &v);
    fq_sub(&t, &v, &x3);
    fq_mul(&y3, &rr, &t);
    fq_mul(&t, F(&p->y), &j);
    fq_dbl(&t, &t);
    fq_sub(&y3, &y3, &t);
    fq_add(&z3, F(&p->z), &h);
    fq_sqr(&z3, &z3);
    fq_sub(&z3, &z3, &z1z1);
    fq_sub(&z3, &z3, &hh);
    *F(&r->x) = x3;
    *F(&r->y) = y3;
    *F(&r->z) = z3;
}
void orc_g1_add(orc_g1_jac *r, const orc_g1_jac *p, const orc_g1_jac *q) {
    if (fq_is_zero(F(&p->z))) { *r = *q; return; }
    if (fq_is_zero(F(&q->z))) { *r = *p; return; }
    fq_t z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t, x3, y3, z3;
    fq_sqr(&z1z1, F(&p->z));
    fq_sqr(&z2z2, F(&q->z));
    fq_mul(&u1, F(&p->x), &z2z2);
    fq_mul(&u2, F(&q->x), &z1z1);
    fq_mul(&s1, F(&p->y), F(&q->z));
    fq_mul(&s1, &s1, &z2z2);
    fq_mul(&s2, F(&q->y), F(&p->z));
    fq_mul(&s2, &s2, &z1z1);
    fq_sub(&h, &u2, &u1);
    fq_sub(&rr, &s2, &s1);
    if (fq_is_zero(&h)) {
        if (fq_is_zero(&rr)) { orc_g1_double(r, p); return; }
        memset(r, 0, sizeof *r);
        fq_one(F(&r->x));
        fq_one(F(&r->y));
        return;
    }
    fq_dbl(&i, &h);
    fq_sqr(&i, &i);
    fq_mul(&j, &h, &i);
    fq_dbl(&rr, &rr);
    fq_mul(&v, &u1, &i);
    fq_sqr(&x3, &rr);
    fq_sub(&x3, &x3, &j);
    fq_sub(&x3, &x3, &v);
    fq_sub(&x3, &x3, &v);
    fq_sub(&t, &v, &x3);
    fq_mul(&y3, &rr, &t);
    fq_mul(&t, &s1, &j);
    fq_dbl(&t, &t);
    fq_sub(&y3, &y3, &t);
    fq_add(&z3, F(&p->z), F(&q->z));
    fq_sqr(&z3, &z3);
    fq_sub(&z3, &z3, &z1z1);
    fq_sub(&z3, &z3, &z2z2);
    fq_mul(&z3, &z3, &h);
    *F(&r->x) = x3;
    *F(&r->y) = y3;
    *F(&r->z) = z3;
}
void orc_g1_to_affine(orc_g1_affine *r, const orc_g1_jac *a) {
    if (fq_is_zero(F(&a->z))) { memset(r, 0, sizeof *r); return; }
    fq_t zi, zi2, zi3;
    fq_inv(&zi, F(&a->z));
    fq_sqr(&zi2, &zi);
    fq_mul(&zi3, &zi2, &zi);
    fq_mul(F(&r->x), F(&a->x), &zi2);
    fq_mul(F(&r->y), F(&a->y), &zi3);
}
void orc_g1_batch_to_affine(orc_g1_affine *r, const orc_g1_jac *a, size_t n) {
    /* Montgomery's trick: one inversion for all non-infinity points */
    fq_t *pre = (fq_t *)malloc(sizeof(fq_t) * (n + 1));
    fq_t acc;
    fq_one(&acc);
    for (size_t i = 0; i < n; i++) {
        pre[i] = acc;
        if (!fq_is_zero(F(&a[i].z))) fq_mul(&acc, &acc, F(&a[i].z));
    }
    fq_inv(&acc, &acc);
    for (size_t i = n; i-- > 0;) {
        if (fq_is_zero(F(&a[i].z))) { memset(&r[i], 0, sizeof r[i]); continue; }
        fq_t zi, zi2, zi3;
        fq_mul(&zi, &acc, &pre[i]);
        fq_mul(&acc, &acc, F(&a[i].z));
        fq_sqr(&zi2, &zi);
        fq_mul(&zi3, &zi2, &zi);
        fq_mul(F(&r[i].x), F(&a[i].x), &zi2);
        fq_mul(F(&r[i].y), F(&a[i].y), &zi3);
    }
    free(pre);
}
void orc_g1_mul(orc_g1_jac *r, const orc_g1_affine *p, const orc_bigint256 *k) {
    /* 4-bit fixed windows from the top set bit: 15 precomputed multiples, one mixed addition per non-zero nibble */
    orc_g1_jac acc, tabj[15];
    orc_g1_affine tab[15], inf;
    memset(&inf, 0, sizeof inf);
    orc_g1_from_affine(&acc, &inf);
    int top = 255;
    while (top >= 0 && !((k->l[top / 64] >> (top % 64)) & 1)) top--;
    if (top < 0 || (fq_is_zero(F(&p->x)) && fq_is_zero(F(&p->y)))) { *r = acc; return; }
    orc_g1_from_affine(&tabj[0], p);
    for (int i = 1; i < 15; i++) orc_g1_add_mixed(&tabj[i], &tabj[i - 1], p);
    orc_g1_batch_to_affine(tab, tabj, 15);
    for (int w = top / 4; w >= 0; w--) {
        for (int d = 0; d < 4; d++) orc_g1_double(&acc, &acc);
        unsigned nib = (unsigned)(k->l[(4 * w) / 64] >> ((4 * w) % 64)) & 15u;
        if (nib) orc_g1_add_mixed(&acc, &acc, &tab[nib - 1]);
    }
    *r = acc;
}

/* fixed-base table for the generator: 32 windows of 8 bits */
static orc_g1_affine *g_fb_table; /* [32][256] */
static void fb_init(void) {
#pragma omp critical(orc_fb_init)
    {
        if (!g_fb_table) {
            orc_g1_affine *tab = (orc_g1_affine *)malloc(sizeof(orc_g1_affine) * 32 * 256);
            orc_g1_jac *row = (orc_g1_jac *)malloc(sizeof(orc_g1_jac) * 256);
            orc_g1_affine base;
            orc_g1_generator(&base);
            for (int w = 0; w < 32; w++) {
                orc_g1_affine inf;
                memset(&inf, 0, sizeof inf);
                orc_g1_from_affine(&row[0], &inf);
                for (int d = 1; d < 256; d++) orc_g1_add_mixed(&row[d], &row[d - 1], &base);
                orc_g1_batch_to_affine(tab + w * 256, row, 256);
                /* base <- 256 * base */
                orc_g1_jac b;
                orc_g1_add_mixed(&b, &row[255], &base);
                orc_g1_to_affine(&base, &b);
            }
            free(row);
            g_fb_table = tab;
        }
    }
}
void orc_g1_fixed_base_batch(orc_g1_affine *out, const orc_fr *k_mont, size_t n) {
    fb_init();
    orc_g1_jac *acc = (orc_g1_jac *)malloc(sizeof(orc_g1_jac) * n);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        uint64_t k[4];
        fr_to_canonical(k, (const fr_t *)&k_mont[i]);
        orc_g1_affine inf;
        memset(&inf, 0, sizeof inf);
        orc_g1_from_affine(&acc[i], &inf);
        for (int w = 0; w < 32; w++) {
            unsigned d = (unsigned)(k[w / 8] >> (8 * (w % 8))) & 0xff;
            if (d) orc_g1_add_mixed(&acc[i], &acc[i], &g_fb_table[w * 256 + d]);
        }
    }
    orc_g1_batch_to_affine(out, acc, n);
    free(acc);
}

/* ------------------------------------------------------------------ G2 (affine, on the D-twist) */
void orc_g2_generator(orc_g2_affine *g) {
    memcpy(&g->x, ORC_G2_GEN_X, 96);
    memcpy(&g->y, ORC_G2_GEN_Y, 96);
}
static int g2_is_inf(const orc_g2_affine *p) { return fq2_is_zero(&p->x) && fq2_is_zero(&p->y); }
int orc_g2_is_on_curve(const orc_g2_affine *p) {
    if (g2_is_inf(p)) return 1;
    orc_fq2 l, r, b;
    memcpy(&b, ORC_TWIST_B, 96);
    fq2_sqr(&l, &p->y);
    fq2_sqr(&r, &p->x);
    fq2_mul(&r, &r, &p->x);
    fq2_add(&r, &r, &b);
    return fq2_eq(&l, &r);
}
/* r = a + b on the twist, also returns the slope; handles doubling; caller excludes infinities/inverses */
static void g2_add_affine(orc_g2_affine *r, orc_fq2 *lambda, const orc_g2_affine *a, const orc_g2_affine *b) {
    orc_fq2 num, den, l, x3, y3, t;
    if (fq2_eq(&a->x, &b->x) && fq2_eq(&a->y, &b->y)) {
        fq2_sqr(&t, &a->x);
        fq2_dbl(&num, &t);
        fq2_add(&num, &num, &t);
        fq2_dbl(&den, &a->y);
    } else {
        fq2_sub(&num, &b->y, &a->y);
        fq2_sub(&den, &b->x, &a->x);
    }
    fq2_inv(&den, &den);
    fq2_mul(&l, &num, &den);
    fq2_sqr(&x3, &l);
    fq2_sub(&x3, &x3, &a->x);
    fq2_sub(&x3, &x3, &b->x);
    fq2_sub(&t, &a->x, &x3);
    fq2_mul(&y3, &l, &t);
    fq2_sub(&y3, &y3, &a->y);
    r->x = x3;
    r->y = y3;
    if (lambda) *lambda = l;
}
void orc_g2_mul(orc_g2_affine *r, const orc_g2_affine *p, const orc_bigint256 *k) {
    orc_g2_affine acc;
    int have = 0;
    memset(&acc, 0, sizeof acc);
    for (int i = 255; i >= 0; i--) {
        if (have) {
            if (fq2_is_zero(&acc.y)) { have = 0; memset(&acc, 0, sizeof acc); }
            else g2_add_affine(&acc, NULL, &acc, &acc);
        }
        if ((k->l[i / 64] >> (i % 64)) & 1) {
            if (!have) { acc = *p; have = !g2_is_inf(p); }
            else {
                orc_fq2 ny;
                fq2_neg(&ny, &p->y);
                if (fq2_eq(&acc.x, &p->x) && fq2_eq(&acc.y, &ny)) { have = 0; memset(&acc, 0, sizeof acc); }
                else g2_add_affine(&acc, NULL, &acc, p);
            }
        }
    }
    *r = acc;
}

/* ------------------------------------------------------------------ pairing */
void orc_g2_prepare(orc_g2_prepared *out, const orc_g2_affine *q) {
    out->n = 0;
    out->infinity = g2_is_inf(q);
    if (out->infinity) return;
    orc_g2_affine T = *q;
    orc_fq2 lam, t;
    for (int i = 62; i >= 0; i--) {
        orc_g2_affine T0 = T;
        g2_add_affine(&T, &lam, &T0, &T0);
        orc_line *ln = &out->lines[out->n++];
        fq2_neg(&ln->c1, &lam);
        fq2_mul(&t, &lam, &T0.x);
        fq2_sub(&ln->c3, &t, &T0.y);
        if ((ORC_BLS_X >> i) & 1) {
            T0 = T;
            g2_add_affine(&T, &lam, &T0, q);
            ln = &out->lines[out->n++];
            fq2_neg(&ln->c1, &lam);
            fq2_mul(&t, &lam, &T0.x);
            fq2_sub(&ln->c3, &t, &T0.y);
        }
    }
}
/* f <- f * prod over the ate loop; f must be initialised by the caller (allows products of pairings) */
static void miller_accumulate(orc_fq12 *f, const orc_g1_affine *const *ps, const orc_g2_prepared *const *qs, size_t n) {
    int idx = 0;
    for (int i = 62; i >= 0; i--) {
        fq12_sqr(f, f);
        for (size_t k = 0; k < n; k++) {
            if (qs[k]->infinity || g1a_is_inf(ps[k])) continue;
            const orc_line *ln = &qs[k]->lines[idx];
            orc_fq2 b;
            fq2_mul_fq(&b, &ln->c1, F(&ps[k]->x));
            fq12_mul_by_line(f, F(&ps[k]->y), &b, &ln->c3);
        }
        idx++;
        if ((ORC_BLS_X >> i) & 1) {
            for (size_t k = 0; k < n; k++) {
                if (qs[k]->infinity || g1a_is_inf(ps[k])) continue;
                const orc_line *ln = &qs[k]->lines[idx];
                orc_fq2 b;
                fq2_mul_fq(&b, &ln->c1, F(&ps[k]->x));
                fq12_mul_by_line(f, F(&ps[k]->y), &b, &ln->c3);
            }
            idx++;
        }
    }
}
void orc_miller_loop(orc_fq12 *f, const orc_g1_affine *p, const orc_g2_prepared *q) {
    orc_fq12_one(f);
    miller_accumulate(f, &p, &q, 1);
}
static void exp_by_x(orc_fq12 *r, const orc_fq12 *a) {
    orc_fq12 acc = *a;
    for (int i = 62; i >= 0; i--) {
        fq12_sqr(&acc, &acc);
        if ((ORC_BLS_X >> i) & 1) orc_fq12_mul(&acc, &acc, a);
    }
    *r = acc;
}
void orc_final_exponentiation(orc_fq12 *r, const orc_fq12 *f) {
    orc_fq12 t, a, b, c, d, e, x;
    /* easy part: f^((q^6-1)(q^2+1)) */
    orc_fq12_inv(&a, f);
    fq12_conj(&t, f);
    orc_fq12_mul(&t, &t, &a);
    orc_fq12_frobenius(&a, &t, 2);
    orc_fq12_mul(&t, &t, &a);
    /* hard part, cubed: 3 (q^4-q^2+1)/r = (x-1)^2 (x+q) (x^2+q^2-1) + 3 */
    exp_by_x(&a, &t);
    fq12_conj(&x, &t);
    orc_fq12_mul(&a, &a, &x); /* t^(x-1) */
    exp_by_x(&b, &a);
    fq12_conj(&x, &a);
    orc_fq12_mul(&b, &b, &x); /* t^((x-1)^2) */
    exp_by_x(&c, &b);
    orc_fq12_frobenius(&x, &b, 1);
    orc_fq12_mul(&c, &c, &x); /* ^(x+q) */
    exp_by_x(&d, &c);
    exp_by_x(&d, &d);
    orc_fq12_frobenius(&x, &c, 2);
    orc_fq12_mul(&d, &d, &x);
    fq12_conj(&x, &c);
    orc_fq12_mul(&d, &d, &x); /* ^(x^2+q^2-1) */
    fq12_sqr(&e, &t);
    orc_fq12_mul(&e, &e, &t);
    orc_fq12_mul(r, &d, &e);
}
void orc_pairing(orc_fq12 *r, const orc_g1_affine *p, const orc_g2_affine *q) {
    orc_g2_prepared *prep = (orc_g2_prepared *)malloc(sizeof *prep);
    orc_fq12 f;
    orc_g2_prepare(prep, q);
    orc_miller_loop(&f, p, prep);
    orc_final_exponentiation(r, &f);
    free(prep);
}
int orc_pairing_product_is_one(const orc_g1_affine *ps, const orc_g2_prepared *const *qs, size_t n) {
    orc_fq12 f, r;
    const orc_g1_affine *pp[8];
    if (n > 8) return -1;
    for (size_t i = 0; i < n; i++) pp[i] = &ps[i];
    orc_fq12_one(&f);
    miller_accumulate(&f, pp, qs, n);
    orc_final_exponentiation(&r, &f);
    return orc_fq12_is_one(&r);
}
