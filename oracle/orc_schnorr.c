/* CPU oracle, Schnorr signatures over Edwards-BLS12.  TEST INFRASTRUCTURE ONLY (see orc.h).
 *
 * What the BFT layer verifies for every signer of a batch certificate: reference node/bft/src/helpers/storage.rs:465
 * (check_certificate), node/bft/src/gateway.rs:1391 (verify_bytes in the handshake), account/src/lib.rs:76-88.  The scheme
 * is snarkVM's console::account::Signature [RECALL; the crate is not in /root/reference: PARITY UNPINNED]; this file
 * follows oracle/pyref.py ("Schnorr signatures over Edwards-BLS12") with different algorithms: Montgomery limbs, extended
 * twisted-Edwards coordinates (add-2008-hwcd-3, dbl-2008-hwcd, a = -1), 4-bit windows. */
#include <stdlib.h>
#include <string.h>
#include <omp.h>
#include "orc_internal.h"

typedef struct { fr_t x, y, z, t; } ed_ext; /* x = X/Z, y = Y/Z, T = XY/Z */

#define ED_W5 5
#define ED_W9 9
#define ED_ROUNDS 39
static struct {
    int ready;
    fr_t d2;              /* 2 d */
    fr_t d;
    orc_ed_affine g;      /* generator of the order-s subgroup */
    fr_t ark5[ED_ROUNDS][ED_W5], mds5[ED_W5][ED_W5];
    fr_t ark9[ED_ROUNDS][ED_W9], mds9[ED_W9][ED_W9];
} K;
static uint64_t ED_ORDER[4]; /* the subgroup order s, set at init from its decimal form */

static void ext_identity(ed_ext *r) { fr_zero(&r->x); fr_one(&r->y); fr_one(&r->z); fr_zero(&r->t); }
static void ext_from_affine(ed_ext *r, const orc_ed_affine *p) {
    r->x = *S(&p->x);
    r->y = *S(&p->y);
    fr_one(&r->z);
    fr_mul(&r->t, &r->x, &r->y);
}
/* unified addition, a = -1 (add-2008-hwcd-3): complete, since d is not a square */
static void ext_add(ed_ext *r, const ed_ext *p, const ed_ext *q) {
    fr_t a, b, c, d, e, f, g, h, t0, t1;
    fr_sub(&t0, &p->y, &p->x); fr_sub(&t1, &q->y, &q->x); fr_mul(&a, &t0, &t1);
    fr_add(&t0, &p->y, &p->x); fr_add(&t1, &q->y, &q->x); fr_mul(&b, &t0, &t1);
    fr_mul(&c, &p->t, &q->t); fr_mul(&c, &c, &K.d2);
    fr_mul(&d, &p->z, &q->z); fr_dbl(&d, &d);
    fr_sub(&e, &b, &a); fr_sub(&f, &d, &c); fr_add(&g, &d, &c); fr_add(&h, &b, &a);
    fr_mul(&r->x, &e, &f); fr_mul(&r->y, &g, &h); fr_mul(&r->t, &e, &h); fr_mul(&r->z, &f, &g);
}
static void ext_dbl(ed_ext *r, const ed_ext *p) { /* dbl-2008-hwcd, a = -1 */
    fr_t a, b, c, d, e, f, g, h, t0;
    fr_sqr(&a, &p->x); fr_sqr(&b, &p->y); fr_sqr(&c, &p->z); fr_dbl(&c, &c);
    fr_neg(&d, &a);
    fr_add(&t0, &p->x, &p->y); fr_sqr(&e, &t0); fr_sub(&e, &e, &a); fr_sub(&e, &e, &b);
    fr_add(&g, &d, &b); fr_sub(&f, &g, &c); fr_sub(&h, &d, &b);
    fr_mul(&r->x, &e, &f); fr_mul(&r->y, &g, &h); fr_mul(&r->t, &e, &h); fr_mul(&r->z, &f, &g);
}
static void ext_mul(ed_ext *r, const ed_ext *p, const uint64_t k[4]) { /* 4-bit windows, k < 2^256 */
    ed_ext tab[16], acc;
    ext_identity(&tab[0]);
    tab[1] = *p;
    for (int i = 2; i < 16; i++) ext_add(&tab[i], &tab[i - 1], p);
    ext_identity(&acc);
    for (int w = 63; w >= 0; w--) {
        for (int j = 0; j < 4; j++) ext_dbl(&acc, &acc);
        const unsigned nib = (unsigned)(k[w / 16] >> (4 * (w % 16))) & 15u;
        if (nib) ext_add(&acc, &acc, &tab[nib]);
    }
    *r = acc;
}
static void ext_to_affine(orc_ed_affine *r, const ed_ext *p) {
    fr_t zi;
    fr_inv(&zi, &p->z);
    fr_mul(S(&r->x), &p->x, &zi);
    fr_mul(S(&r->y), &p->y, &zi);
}
static int ext_eq_affine(const ed_ext *p, const orc_ed_affine *q) { /* X == x Z and Y == y Z */
    fr_t t;
    fr_mul(&t, S(&q->x), &p->z);
    if (!fr_eq(&t, &p->x)) return 0;
    fr_mul(&t, S(&q->y), &p->z);
    return fr_eq(&t, &p->y);
}
int orc_ed_on_curve(const orc_ed_affine *p) { /* -x^2 + y^2 == 1 + d x^2 y^2, coordinates canonical-range Montgomery */
    fr_t xx, yy, l, r, one;
    for (int i = 3; i >= 0; i--) {
        if (p->x.l[i] < ORC_FR_MODULUS[i]) break;
        if (p->x.l[i] > ORC_FR_MODULUS[i] || i == 0) return 0;
    }
    for (int i = 3; i >= 0; i--) {
        if (p->y.l[i] < ORC_FR_MODULUS[i]) break;
        if (p->y.l[i] > ORC_FR_MODULUS[i] || i == 0) return 0;
    }
    fr_one(&one);
    fr_sqr(&xx, S(&p->x)); fr_sqr(&yy, S(&p->y));
    fr_sub(&l, &yy, &xx);
    fr_mul(&r, &xx, &yy); fr_mul(&r, &r, &K.d); fr_add(&r, &r, &one);
    return fr_eq(&l, &r);
}

/* Tonelli-Shanks in Fr (two-adicity 47, non-residue 22); init-time only */
static int fr_sqrt_slow(fr_t *out, const fr_t *v) {
    fr_t one, w, x, b, c, g;
    fr_one(&one);
    if (fr_is_zero(v)) { *out = *v; return 1; }
    uint64_t t[4], e[4];
    memcpy(t, ORC_FR_MODULUS, 32);
    t[0] -= 1;
    for (int s = 0; s < 47; s++) { /* t = (r - 1) >> 47 */
        for (int i = 0; i < 4; i++) t[i] = (t[i] >> 1) | (i < 3 ? t[i + 1] << 63 : 0);
    }
    memcpy(e, t, 32); /* (t - 1) / 2 */
    for (int i = 0; i < 4; i++) e[i] = (e[i] >> 1) | (i < 3 ? e[i + 1] << 63 : 0);
    fr_pow(&w, v, e, 4);
    fr_mul(&x, v, &w);
    fr_mul(&b, &x, &w);
    fr_t gen;
    fr_from_u64(&gen, 22);
    fr_pow(&c, &gen, t, 4);
    int m = 47;
    while (!fr_eq(&b, &one)) {
        int i = 0;
        fr_t b2 = b;
        while (!fr_eq(&b2, &one)) {
            fr_sqr(&b2, &b2);
            if (++i == m) return 0;
        }
        g = c;
        for (int k = 0; k < m - i - 1; k++) fr_sqr(&g, &g);
        fr_mul(&x, &x, &g);
        fr_sqr(&c, &g);
        fr_mul(&b, &b, &c);
        m = i;
    }
    *out = x;
    return 1;
}
static int canon_less(const fr_t *a, const fr_t *b) {
    uint64_t x[4], y[4];
    fr_to_canonical(x, a);
    fr_to_canonical(y, b);
    for (int i = 3; i >= 0; i--) {
        if (x[i] < y[i]) return 1;
        if (x[i] > y[i]) return 0;
    }
    return 0;
}
static void fr_from_bytes64(fr_t *r, const uint8_t *b) { /* 512-bit little-endian integer mod r (Horner over bytes) */
    fr_t acc, c256, t;
    fr_zero(&acc);
    fr_from_u64(&c256, 256);
    for (int i = 63; i >= 0; i--) {
        fr_mul(&acc, &acc, &c256);
        fr_from_u64(&t, b[i]);
        fr_add(&acc, &acc, &t);
    }
    *r = acc;
}
static void ed_init(void) {
#pragma omp critical(orc_ed_init)
    if (!K.ready) {
        const char *dec = "2111115437357092606062206234695386632838870926408408195193685246394721360383";
        uint64_t v[4] = {0, 0, 0, 0};
        for (const char *q = dec; *q; q++) {
            unsigned __int128 carry = (unsigned)(*q - '0');
            for (int i = 0; i < 4; i++) {
                unsigned __int128 w = (unsigned __int128)v[i] * 10u + carry;
                v[i] = (uint64_t)w;
                carry = w >> 64;
            }
        }
        memcpy(ED_ORDER, v, 32);
        fr_from_u64(&K.d, 3021);
        fr_dbl(&K.d2, &K.d);
        static const char TAG[] = "MI355X-Poseidon-BLS12-377-Fr-v1";
        for (int w = 5; w <= 9; w += 4) {
            for (int r = 0; r < ED_ROUNDS; r++)
                for (int i = 0; i < w; i++) {
                    uint8_t seed[64], h[64];
                    const size_t n = sizeof TAG - 1;
                    memcpy(seed, TAG, n);
                    seed[n] = (uint8_t)w; seed[n + 1] = (uint8_t)r; seed[n + 2] = (uint8_t)i;
                    seed[n + 3] = 0;
                    orc_sha256(h, seed, n + 4);
                    seed[n + 3] = 1;
                    orc_sha256(h + 32, seed, n + 4);
                    fr_from_bytes64(w == 5 ? &K.ark5[r][i] : &K.ark9[r][i], h);
                }
            for (int i = 0; i < w; i++)
                for (int j = 0; j < w; j++) {
                    fr_t t;
                    fr_from_u64(&t, (uint64_t)(i + w + j));
                    fr_inv(w == 5 ? &K.mds5[i][j] : &K.mds9[i][j], &t);
                }
        }
        /* generator: 4 (x, y), smallest x >= 2 on the curve, y the smaller root */
        fr_t one;
        fr_one(&one);
        for (uint64_t xv = 2;; xv++) {
            fr_t x, xx, num, den, y, ny;
            fr_from_u64(&x, xv);
            fr_sqr(&xx, &x);
            fr_add(&num, &one, &xx);                          /* 1 - a x^2 = 1 + x^2 */
            fr_mul(&den, &K.d, &xx); fr_sub(&den, &one, &den); /* 1 - d x^2 */
            fr_inv(&den, &den);
            fr_mul(&num, &num, &den);
            if (!fr_sqrt_slow(&y, &num)) continue;
            fr_neg(&ny, &y);
            if (canon_less(&ny, &y)) y = ny;
            orc_ed_affine p;
            *S(&p.x) = x;
            *S(&p.y) = y;
            ed_ext e;
            ext_from_affine(&e, &p);
            ext_dbl(&e, &e);
            ext_dbl(&e, &e);
            if (fr_is_zero(&e.x)) continue; /* identity (or the point of order two) */
            ext_to_affine(&K.g, &e);
            break;
        }
        K.ready = 1;
    }
}
void orc_ed_generator(orc_ed_affine *g) { ed_init(); *g = K.g; }
void orc_ed_order(uint64_t s[4]) { ed_init(); memcpy(s, ED_ORDER, 32); }
void orc_ed_mul(orc_ed_affine *r, const orc_ed_affine *p, const uint64_t k[4]) {
    ed_init();
    ed_ext e;
    ext_from_affine(&e, p);
    ext_mul(&e, &e, k);
    ext_to_affine(r, &e);
}
void orc_ed_add(orc_ed_affine *r, const orc_ed_affine *p, const orc_ed_affine *q) {
    ed_init();
    ed_ext a, b;
    ext_from_affine(&a, p);
    ext_from_affine(&b, q);
    ext_add(&a, &a, &b);
    ext_to_affine(r, &a);
}

static void fr_pow17(fr_t *r, const fr_t *x) {
    fr_t x2, x4, x8, x16;
    fr_sqr(&x2, x); fr_sqr(&x4, &x2); fr_sqr(&x8, &x4); fr_sqr(&x16, &x8);
    fr_mul(r, &x16, x);
}
void orc_ed_poseidon_permute(orc_fr *st, int width) {
    ed_init();
    fr_t *s = (fr_t *)st, nx[ED_W9];
    for (int r = 0; r < ED_ROUNDS; r++) {
        for (int i = 0; i < width; i++) fr_add(&s[i], &s[i], width == 5 ? &K.ark5[r][i] : &K.ark9[r][i]);
        if (r < 4 || r >= 4 + 31) for (int i = 0; i < width; i++) fr_pow17(&s[i], &s[i]);
        else fr_pow17(&s[0], &s[0]);
        for (int i = 0; i < width; i++) {
            fr_zero(&nx[i]);
            for (int j = 0; j < width; j++) {
                fr_t t;
                fr_mul(&t, width == 5 ? &K.mds5[i][j] : &K.mds9[i][j], &s[j]);
                fr_add(&nx[i], &nx[i], &t);
            }
        }
        memcpy(s, nx, sizeof(fr_t) * (size_t)width);
    }
}
/* one permutation: capacity = rate 2^64 + n, inputs in the rate part; low 250 bits of the first rate element (canonical) */
void orc_ed_hash_to_scalar(uint64_t out[4], const orc_fr *inputs, int n, int rate) {
    fr_t st[ED_W9];
    const uint64_t cap[4] = {(uint64_t)n, (uint64_t)rate, 0, 0};
    fr_from_canonical(&st[0], cap);
    for (int i = 0; i < rate; i++) {
        if (i < n) st[1 + i] = *(const fr_t *)&inputs[i];
        else fr_zero(&st[1 + i]);
    }
    orc_ed_poseidon_permute((orc_fr *)st, rate + 1);
    fr_to_canonical(out, &st[1]);
    out[3] &= 0x03ffffffffffffffULL; /* 250 bits */
}
static int scalar_lt_order(const uint64_t k[4]) {
    for (int i = 3; i >= 0; i--) {
        if (k[i] < ED_ORDER[i]) return 1;
        if (k[i] > ED_ORDER[i]) return 0;
    }
    return 0;
}
int orc_schnorr_verify_one(const orc_sig_record *r) {
    ed_init();
    if (!scalar_lt_order(r->challenge) || !scalar_lt_order(r->response)) return 0;
    if (!orc_ed_on_curve(&r->pk_sig) || !orc_ed_on_curve(&r->pr_sig) || !orc_ed_on_curve(&r->address)) return 0;
    {
        uint64_t m[4];
        memcpy(m, &r->message, 32);
        for (int i = 3; i >= 0; i--) {
            if (m[i] < ORC_FR_MODULUS[i]) break;
            if (m[i] > ORC_FR_MODULUS[i] || i == 0) return 0;
        }
    }
    ed_ext g, pk, a, b;
    ext_from_affine(&g, &K.g);
    ext_from_affine(&pk, &r->pk_sig);
    ext_mul(&a, &g, r->response);
    ext_mul(&b, &pk, r->challenge);
    ext_add(&a, &a, &b);
    orc_ed_affine gr;
    ext_to_affine(&gr, &a);
    orc_fr in[5] = {gr.x, r->pk_sig.x, r->pr_sig.x, r->address.x, r->message};
    uint64_t c[4];
    orc_ed_hash_to_scalar(c, in, 5, 8);
    if (memcmp(c, r->challenge, 32)) return 0;
    orc_fr in2[2] = {r->pk_sig.x, r->pr_sig.x};
    uint64_t prf[4];
    orc_ed_hash_to_scalar(prf, in2, 2, 4);
    ed_ext pr;
    ext_from_affine(&pr, &r->pr_sig);
    ext_mul(&a, &g, prf);
    ext_add(&a, &a, &pk);
    ext_add(&a, &a, &pr);
    return ext_eq_affine(&a, &r->address);
}
void orc_schnorr_verify_batch(const orc_sig_record *recs, size_t n, uint8_t *verdicts, int threads) {
    ed_init();
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads)
    for (size_t i = 0; i < n; i++) verdicts[i] = (uint8_t)orc_schnorr_verify_one(&recs[i]);
}
/* synthetic certificates: n_signers accounts (keys from the seed), every record signs message[i / n_signers]... here:
 * record i is signer i % n_signers over message i / n_signers (a batch id per certificate); corrupt_per_1024 of 1024 are spoilt */
static uint64_t smx(uint64_t *s) {
    *s += 0x9E3779B97F4A7C15ull;
    uint64_t z = *s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static void smx_scalar(uint64_t *s, uint64_t k[4]) { /* below 2^250 < s */
    for (int i = 0; i < 4; i++) k[i] = smx(s);
    k[3] &= 0x03ffffffffffffffULL;
}
static void scalar_mulsub(uint64_t out[4], const uint64_t k[4], const uint64_t c[4], const uint64_t sk[4]) {
    /* (k - c sk) mod s with 512-bit schoolbook arithmetic and reduction by repeated conditional subtraction of shifted s */
    unsigned __int128 acc;
    uint64_t prod[8] = {0};
    for (int i = 0; i < 4; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < 4; j++) {
            acc = (unsigned __int128)c[i] * sk[j] + prod[i + j] + carry;
            prod[i + j] = (uint64_t)acc;
            carry = (uint64_t)(acc >> 64);
        }
        prod[i + 4] = carry;
    }
    /* prod mod s: binary long division */
    uint64_t rem[5] = {0};
    for (int bit = 511; bit >= 0; bit--) {
        for (int i = 4; i > 0; i--) rem[i] = (rem[i] << 1) | (rem[i - 1] >> 63);
        rem[0] = (rem[0] << 1) | ((prod[bit / 64] >> (bit % 64)) & 1);
        int ge = rem[4] != 0;
        if (!ge) {
            ge = 1;
            for (int i = 3; i >= 0; i--) {
                if (rem[i] > ED_ORDER[i]) break;
                if (rem[i] < ED_ORDER[i]) { ge = 0; break; }
            }
        }
        if (ge) {
            uint64_t borrow = 0;
            for (int i = 0; i < 4; i++) {
                unsigned __int128 d = (unsigned __int128)rem[i] - ED_ORDER[i] - borrow;
                rem[i] = (uint64_t)d;
                borrow = (uint64_t)(d >> 64) & 1;
            }
            rem[4] -= borrow;
        }
    }
    /* out = k - rem mod s (k < s, rem < s) */
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        unsigned __int128 d = (unsigned __int128)k[i] - rem[i] - borrow;
        out[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    if (borrow) {
        uint64_t carry = 0;
        for (int i = 0; i < 4; i++) {
            unsigned __int128 d = (unsigned __int128)out[i] + ED_ORDER[i] + carry;
            out[i] = (uint64_t)d;
            carry = (uint64_t)(d >> 64);
        }
    }
}
void orc_schnorr_generate(uint64_t seed, size_t n, size_t n_signers, uint32_t corrupt_per_1024, orc_sig_record *recs, uint8_t *expected) {
    ed_init();
    uint64_t st = seed;
    ed_ext g;
    ext_from_affine(&g, &K.g);
    uint64_t (*sk)[4] = malloc(sizeof(uint64_t[4]) * n_signers);
    orc_ed_affine *pk = malloc(sizeof(orc_ed_affine) * n_signers), *pr = malloc(sizeof(orc_ed_affine) * n_signers),
                  *ad = malloc(sizeof(orc_ed_affine) * n_signers);
    for (size_t a = 0; a < n_signers; a++) {
        uint64_t rs[4], prf[4];
        smx_scalar(&st, sk[a]);
        smx_scalar(&st, rs);
        ed_ext e, f;
        ext_mul(&e, &g, sk[a]); ext_to_affine(&pk[a], &e);
        ext_mul(&f, &g, rs); ext_to_affine(&pr[a], &f);
        orc_fr in2[2] = {pk[a].x, pr[a].x};
        orc_ed_hash_to_scalar(prf, in2, 2, 4);
        ed_ext h;
        ext_mul(&h, &g, prf);
        ext_add(&h, &h, &e);
        ext_add(&h, &h, &f);
        ext_to_affine(&ad[a], &h);
    }
    for (size_t i = 0; i < n; i++) {
        orc_sig_record *r = &recs[i];
        const size_t a = i % n_signers;
        memset(r, 0, sizeof *r);
        uint64_t m[4] = {0x1000 + i / n_signers, seed, 0, 0}, k[4];
        fr_from_canonical(S(&r->message), m);
        r->address = ad[a];
        r->pk_sig = pk[a];
        r->pr_sig = pr[a];
        smx_scalar(&st, k);
        ed_ext e;
        orc_ed_affine gr;
        ext_mul(&e, &g, k);
        ext_to_affine(&gr, &e);
        orc_fr in[5] = {gr.x, r->pk_sig.x, r->pr_sig.x, r->address.x, r->message};
        orc_ed_hash_to_scalar(r->challenge, in, 5, 8);
        scalar_mulsub(r->response, k, r->challenge, sk[a]);
        expected[i] = 1;
    }
    for (size_t i = 0; i < n; i++) {
        if ((smx(&st) & 1023) >= corrupt_per_1024) continue;
        orc_sig_record *r = &recs[i];
        switch ((int)(smx(&st) % 5)) {
        case 0: r->response[0] ^= 1; break;
        case 1: r->challenge[1] ^= 4; break;
        case 2: { fr_t one; fr_one(&one); fr_add(S(&r->message), S(&r->message), &one); break; }
        case 3: r->pr_sig = K.g; break;                 /* the address identity breaks (and the challenge) */
        default: r->address = recs[(i + 1) % n].address; if (n_signers == 1) r->address = K.g; break;
        }
        expected[i] = 0;
    }
    free(sk); free(pk); free(pr); free(ad);
}
