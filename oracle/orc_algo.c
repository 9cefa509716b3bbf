/* CPU oracle: MSM, NTT, KZG10, SHA-256, VarunaShaped-v1.  TEST INFRASTRUCTURE ONLY (see orc.h).
 *
 * Restates (published algorithms; snarkVM 1.0.0 @ dea322b sources are not available):
 *   msm::VariableBase::msm            -> orc_msm_naive / orc_msm_pippenger   
 *   fft::EvaluationDomain::*_in_place -> orc_ntt                             
 *   kzg10::KZG10::{check,batch_check} -> orc_kzg_check / orc_kzg_batch_check
 *   the per-transaction verify loop of VM::check_transactions (rayon over transactions, one
 *   pairing product per proof)        -> orc_vs_verify_batch_cpu             
 */
#include "orc_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static void g1_set_inf(orc_g1_jac *r) {
    memset(r, 0, sizeof *r);
    fq_one(F(&r->x));
    fq_one(F(&r->y));
}

/* ------------------------------------------------------------------ MSM */
void orc_msm_naive(orc_g1_jac *out, const orc_g1_affine *bases, const orc_bigint256 *scalars, size_t n) {
    orc_g1_jac acc, t;
    g1_set_inf(&acc);
    for (size_t i = 0; i < n; i++) {
        orc_g1_mul(&t, &bases[i], &scalars[i]);
        orc_g1_add(&acc, &acc, &t);
    }
    *out = acc;
}

static inline unsigned get_window(const orc_bigint256 *s, int bit, int c) {
    int limb = bit >> 6, off = bit & 63;
    uint64_t v = s->l[limb] >> off;
    if (off + c > 64 && limb < 3) v |= s->l[limb + 1] << (64 - off);
    return (unsigned)(v & ((1ull << c) - 1));
}

void orc_msm_pippenger(orc_g1_jac *out, const orc_g1_affine *bases, const orc_bigint256 *scalars, size_t n,
                       int threads) {
    if (n < 15) { orc_msm_naive(out, bases, scalars, n); return; }
    int c = (int)floor(log((double)n)) + 2; /* snarkVM's window heuristic */
    if (c > 20) c = 20;
    int nwin = (253 + c - 1) / c;
    orc_g1_jac *wsum = (orc_g1_jac *)malloc(sizeof(orc_g1_jac) * nwin);
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int w = 0; w < nwin; w++) {
        size_t nb = ((size_t)1 << c) - 1;
        orc_g1_jac *buckets = (orc_g1_jac *)malloc(sizeof(orc_g1_jac) * nb);
        for (size_t b = 0; b < nb; b++) g1_set_inf(&buckets[b]);
        for (size_t i = 0; i < n; i++) {
            unsigned d = get_window(&scalars[i], w * c, c);
            if (d) orc_g1_add_mixed(&buckets[d - 1], &buckets[d - 1], &bases[i]);
        }
        orc_g1_jac run, sum;
        g1_set_inf(&run);
        g1_set_inf(&sum);
        for (size_t b = nb; b-- > 0;) {
            orc_g1_add(&run, &run, &buckets[b]);
            orc_g1_add(&sum, &sum, &run);
        }
        wsum[w] = sum;
        free(buckets);
    }
    orc_g1_jac acc;
    g1_set_inf(&acc);
    for (int w = nwin - 1; w >= 0; w--) {
        for (int k = 0; k < c; k++) orc_g1_double(&acc, &acc);
        orc_g1_add(&acc, &acc, &wsum[w]);
    }
    free(wsum);
    *out = acc;
}

/* ------------------------------------------------------------------ NTT */
void orc_fr_omega(orc_fr *w, uint32_t log_n) {
    fr_t t;
    memcpy(&t, ORC_FR_ROOT_OF_UNITY, 32);
    for (uint32_t i = log_n; i < ORC_FR_TWO_ADICITY; i++) fr_sqr(&t, &t);
    memcpy(w, &t, 32);
}
void orc_dft_naive(orc_fr *out, const orc_fr *in, uint32_t log_n) {
    size_t n = (size_t)1 << log_n;
    fr_t w, wi, wij, acc, t;
    orc_fr_omega((orc_fr *)&w, log_n);
    fr_one(&wi);
    for (size_t i = 0; i < n; i++) {
        fr_zero(&acc);
        fr_one(&wij);
        for (size_t j = 0; j < n; j++) {
            fr_mul(&t, S(&in[j]), &wij);
            fr_add(&acc, &acc, &t);
            fr_mul(&wij, &wij, &wi);
        }
        memcpy(&out[i], &acc, 32);
        fr_mul(&wi, &wi, &w);
    }
}
static size_t bitrev(size_t x, uint32_t bits) {
    size_t r = 0;
    for (uint32_t i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}
int orc_ntt(orc_fr *data, uint32_t log_n, int inverse, int coset, int threads) {
    if (log_n > ORC_FR_TWO_ADICITY) return -1;
    size_t n = (size_t)1 << log_n;
    fr_t *a = (fr_t *)data;
    fr_t w, g, gi;
    if (threads < 1) threads = 1;
    orc_fr_omega((orc_fr *)&w, log_n);
    if (inverse) fr_inv(&w, &w);
    memcpy(&g, ORC_FR_GENERATOR, 32);
    memcpy(&gi, ORC_FR_GENERATOR_INV, 32);
    if (coset && !inverse) {
        fr_t p;
        fr_one(&p);
        for (size_t i = 0; i < n; i++) { fr_mul(&a[i], &a[i], &p); fr_mul(&p, &p, &g); }
    }
    for (size_t i = 0; i < n; i++) {
        size_t j = bitrev(i, log_n);
        if (i < j) { fr_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    /* twiddle table w^k, k < n/2 */
    size_t half = n / 2 ? n / 2 : 1;
    fr_t *tw = (fr_t *)malloc(sizeof(fr_t) * half);
    fr_one(&tw[0]);
    for (size_t k = 1; k < half; k++) fr_mul(&tw[k], &tw[k - 1], &w);
    for (uint32_t s = 1; s <= log_n; s++) {
        size_t m = (size_t)1 << s, hm = m >> 1, stride = n / m;
#pragma omp parallel for schedule(static) num_threads(threads) if (n >= 4096)
        for (size_t idx = 0; idx < n / 2; idx++) {
            size_t blk = idx / hm, j = idx % hm;
            size_t i0 = blk * m + j, i1 = i0 + hm;
            fr_t t, u = a[i0];
            fr_mul(&t, &a[i1], &tw[j * stride]);
            fr_add(&a[i0], &u, &t);
            fr_sub(&a[i1], &u, &t);
        }
    }
    free(tw);
    if (inverse) {
        fr_t ninv, p;
        fr_from_u64(&ninv, (uint64_t)n);
        fr_inv(&ninv, &ninv);
        if (coset) {
            fr_one(&p);
            for (size_t i = 0; i < n; i++) {
                fr_mul(&a[i], &a[i], &ninv);
                fr_mul(&a[i], &a[i], &p);
                fr_mul(&p, &p, &gi);
            }
        } else {
#pragma omp parallel for schedule(static) num_threads(threads) if (n >= 4096)
            for (size_t i = 0; i < n; i++) fr_mul(&a[i], &a[i], &ninv);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ KZG10 */
static void fr_from_u128le(fr_t *r, const uint8_t *b16) {
    uint64_t l[4] = {0, 0, 0, 0};
    memcpy(l, b16, 16);
    fr_from_canonical(r, l);
}
static void g1_mul_fr(orc_g1_jac *r, const orc_g1_affine *p, const fr_t *k_mont) {
    orc_bigint256 k;
    fr_to_canonical(k.l, k_mont);
    orc_g1_mul(r, p, &k);
}
void orc_kzg_commit(orc_g1_jac *out, const orc_g1_affine *powers, const orc_fr *coeffs, size_t n, int threads) {
    orc_bigint256 *plain = (orc_bigint256 *)malloc(sizeof(orc_bigint256) * (n ? n : 1));
    for (size_t i = 0; i < n; i++) orc_fr_to_canonical(&plain[i], &coeffs[i]);
    orc_msm_pippenger(out, powers, plain, n, threads);
    free(plain);
}
void orc_poly_mul_schoolbook(orc_fr *out, const orc_fr *a, size_t na, const orc_fr *b, size_t nb) {
    fr_t t;
    for (size_t k = 0; k < na + nb - 1; k++) fr_zero((fr_t *)&out[k]);
    for (size_t i = 0; i < na; i++)
        for (size_t j = 0; j < nb; j++) {
            fr_mul(&t, S(&a[i]), S(&b[j]));
            fr_add((fr_t *)&out[i + j], (fr_t *)&out[i + j], &t);
        }
}
void orc_poly_eval(orc_fr *out, const orc_fr *coeffs, size_t n, const orc_fr *z) {
    fr_t acc;
    fr_zero(&acc);
    for (size_t i = n; i-- > 0;) {
        fr_mul(&acc, &acc, S(z));
        fr_add(&acc, &acc, S(&coeffs[i]));
    }
    *(fr_t *)out = acc;
}
void orc_poly_witness(orc_fr *q, orc_fr *value, const orc_fr *coeffs, size_t n, const orc_fr *z) {
    /* p(x) = (x - z) q(x) + p(z):  q_{n-2} = c_{n-1},  q_{i-1} = c_i + z q_i,  p(z) = c_0 + z q_0 */
    fr_t carry;
    fr_zero(&carry);
    for (size_t i = n; i-- > 0;) {
        fr_t t;
        fr_mul(&t, &carry, S(z));
        fr_add(&t, &t, S(&coeffs[i]));
        if (i > 0) *(fr_t *)&q[i - 1] = t;
        else *(fr_t *)value = t;
        carry = t;
    }
}
int orc_kzg_check(const orc_kzg_vk *vk, const orc_g1_affine *comm, const orc_fr *point, const orc_fr *value,
                  const orc_g1_affine *witness) {
    uint8_t one[16] = {1};
    return orc_kzg_batch_check(vk, comm, point, value, witness, 1, one);
}
int orc_kzg_batch_check(const orc_kzg_vk *vk, const orc_g1_affine *comms, const orc_fr *points,
                        const orc_fr *values, const orc_g1_affine *witnesses, size_t n,
                        const uint8_t *randomizers128) {
    return orc_kzg_batch_check_hiding(vk, NULL, comms, points, values, witnesses, NULL, n, randomizers128);
}
/* kzg10::KZG10::batch_check with hiding: VerifierKey.gamma_g and KZGProof.random_v,
 *   total_c = sum r_i (C_i + z_i w_i) - (sum r_i v_i) g - (sum r_i rbar_i) gamma_g,  total_w = sum r_i w_i,
 *   e(total_c, h) e(-total_w, beta_h) == 1.   gamma_g / random_v NULL: no hiding. */
int orc_kzg_batch_check_hiding(const orc_kzg_vk *vk, const orc_g1_affine *gamma_g, const orc_g1_affine *comms,
                               const orc_fr *points, const orc_fr *values, const orc_g1_affine *witnesses,
                               const orc_fr *random_v, size_t n, const uint8_t *randomizers128) {
    orc_g1_jac total_c, total_w, t;
    fr_t gs, hs, r, rz, rv;
    g1_set_inf(&total_c);
    g1_set_inf(&total_w);
    fr_zero(&gs);
    fr_zero(&hs);
    for (size_t i = 0; i < n; i++) {
        fr_from_u128le(&r, randomizers128 + 16 * i);
        g1_mul_fr(&t, &comms[i], &r);
        orc_g1_add(&total_c, &total_c, &t);
        fr_mul(&rz, &r, S(&points[i]));
        g1_mul_fr(&t, &witnesses[i], &rz);
        orc_g1_add(&total_c, &total_c, &t);
        g1_mul_fr(&t, &witnesses[i], &r);
        orc_g1_add(&total_w, &total_w, &t);
        fr_mul(&rv, &r, S(&values[i]));
        fr_add(&gs, &gs, &rv);
        if (gamma_g && random_v) {
            fr_mul(&rv, &r, S(&random_v[i]));
            fr_add(&hs, &hs, &rv);
        }
    }
    fr_neg(&gs, &gs);
    g1_mul_fr(&t, &vk->g, &gs);
    orc_g1_add(&total_c, &total_c, &t);
    if (gamma_g && random_v) {
        fr_neg(&hs, &hs);
        g1_mul_fr(&t, gamma_g, &hs);
        orc_g1_add(&total_c, &total_c, &t);
    }
    orc_g1_affine ps[2];
    orc_g1_to_affine(&ps[0], &total_c);
    orc_g1_to_affine(&ps[1], &total_w);
    orc_g1_neg_affine(&ps[1], &ps[1]);
    orc_g2_prepared *h = (orc_g2_prepared *)malloc(sizeof *h), *th = (orc_g2_prepared *)malloc(sizeof *th);
    orc_g2_prepare(h, &vk->h);
    orc_g2_prepare(th, &vk->tau_h);
    const orc_g2_prepared *qs[2] = {h, th};
    int ok = orc_pairing_product_is_one(ps, qs, 2);
    free(h);
    free(th);
    return ok;
}

/* ------------------------------------------------------------------ Poseidon over Fq */
static void fq_pow17(fq_t *r, const fq_t *x) {
    fq_t x2, x4, x8, x16;
    fq_sqr(&x2, x);
    fq_sqr(&x4, &x2);
    fq_sqr(&x8, &x4);
    fq_sqr(&x16, &x8);
    fq_mul(r, &x16, x);
}
void orc_poseidon_permute(orc_fq st[3]) {
    for (int r = 0; r < ORC_POSEIDON_ROUNDS; r++) {
        for (int i = 0; i < 3; i++) fq_add(&st[i], &st[i], (const fq_t *)ORC_POSEIDON_ARK[r][i]);
        if (r < ORC_POSEIDON_FULL_HALF || r >= ORC_POSEIDON_FULL_HALF + ORC_POSEIDON_PARTIAL) {
            for (int i = 0; i < 3; i++) fq_pow17(&st[i], &st[i]);
        } else {
            fq_pow17(&st[0], &st[0]);
        }
        fq_t o[3], t;
        for (int i = 0; i < 3; i++) {
            fq_zero(&o[i]);
            for (int j = 0; j < 3; j++) {
                fq_mul(&t, (const fq_t *)ORC_POSEIDON_MDS[i][j], &st[j]);
                fq_add(&o[i], &o[i], &t);
            }
        }
        for (int i = 0; i < 3; i++) st[i] = o[i];
    }
}
void orc_poseidon_hash_many(const orc_fq *inputs, size_t n, size_t m, size_t k, orc_fq *outputs, int threads) {
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (size_t s = 0; s < n; s++) {
        orc_fq st[3];
        memset(st, 0, sizeof st);
        const orc_fq *in = inputs + s * m;
        for (size_t i = 0; i < m; i += 2) {
            fq_add(&st[1], &st[1], &in[i]);
            if (i + 1 < m) fq_add(&st[2], &st[2], &in[i + 1]);
            orc_poseidon_permute(st);
        }
        if (m == 0) orc_poseidon_permute(st);
        size_t produced = 0;
        while (produced < k) {
            outputs[s * k + produced++] = st[1];
            if (produced < k) outputs[s * k + produced++] = st[2];
            if (produced < k) orc_poseidon_permute(st);
        }
    }
}

/* ------------------------------------------------------------------ SHA-256 (FIPS 180-4) */
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98,
    0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786,
    0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8,
    0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819,
    0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a,
    0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7,
    0xc67178f2};
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha256_block(uint32_t st[8], const uint8_t *p) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
        w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    for (int i = 0; i < 64; i++) {
        uint32_t t1 = h + (ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
        uint32_t t2 = (ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}
void orc_sha256(uint8_t out[32], const uint8_t *data, size_t len) {
    uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    size_t i = 0;
    for (; i + 64 <= len; i += 64) sha256_block(st, data + i);
    uint8_t buf[128];
    size_t rem = len - i;
    memset(buf, 0, sizeof buf);
    memcpy(buf, data + i, rem);
    buf[rem] = 0x80;
    size_t tot = rem + 9 <= 64 ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int k = 0; k < 8; k++) buf[tot - 1 - k] = (uint8_t)(bits >> (8 * k));
    sha256_block(st, buf);
    if (tot == 128) sha256_block(st, buf + 64);
    for (int k = 0; k < 8; k++) {
        out[4 * k] = (uint8_t)(st[k] >> 24);
        out[4 * k + 1] = (uint8_t)(st[k] >> 16);
        out[4 * k + 2] = (uint8_t)(st[k] >> 8);
        out[4 * k + 3] = (uint8_t)st[k];
    }
}

/* ------------------------------------------------------------------ VarunaShaped-v1 */
static const char VS_TAG[32] = "MI355X-VarunaShaped-v1";

/* Fiat-Shamir kind of the synthetic shape: 0 = SHA-256 over the in-memory bytes, 1 = Poseidon sponge over Fq with the
 * reference's absorb volume (37 permutations per proof).  Process-wide switch (test infrastructure: set it before a
 * run, never two kinds at once). */
static int g_vs_fs_kind = 0;
void orc_vs_set_transcript(int kind) { g_vs_fs_kind = kind ? 1 : 0; }
int orc_vs_get_transcript(void) { return g_vs_fs_kind; }

/* bytes (<= 48) -> the Fq element whose Montgomery limbs are those bytes, zero-extended */
static void fs_embed(fq_t *r, const void *b, size_t n) {
    memset(r, 0, sizeof *r);
    memcpy(r, b, n);
}
/* the e-th of the 73 + 38 m1 absorbed elements (oracle/pyref.py: vs_fs_elements); m1 further instances (768-byte
 * records: 16 inputs, two commitments, two evaluations) sit after C_1 (36 elements each) and at the end (2 each) */
/* VarunaShaped-v3: d1 further circuits (parts: 400-byte records {id, first instance, pad, 12 evaluations}) put their
 * header and digest after the first circuit's digest (2 elements each) and their evaluations at the very end (12 each) */
typedef struct { const orc_vs_circuit_part *parts; int d1; const orc_vs_circuit_vk *circuits; } vs_parts;
static const vs_parts VS_NO_PARTS = {NULL, 0, NULL};
static void fs_element(fq_t *r, const orc_vs_circuit_vk *cvk, const orc_vs_proof *p, const uint8_t *extra, int m1, const vs_parts *pt, int e) {
    const uint8_t *comms = (const uint8_t *)p->comms;
    if (e == 0) { fs_embed(r, VS_TAG, 32); return; }
    if (e == 1) { fs_embed(r, cvk->digest, 32); return; }
    if (e < 2 + 2 * pt->d1) {
        const orc_vs_circuit_part *cp = &pt->parts[(e - 2) >> 1];
        if ((e - 2) & 1) fs_embed(r, pt->circuits[cp->circuit].digest, 32);
        else fs_embed(r, cp, 16);
        return;
    }
    e -= 2 * pt->d1;
    if (e < 34) { fs_embed(r, (const uint8_t *)&p->inputs[(e - 2) >> 1] + 16 * ((e - 2) & 1), 16); return; }
    if (e < 38) { fs_embed(r, comms + 48 * (e - 34), 48); return; }
    e -= 38;
    if (e < 36 * m1) {
        const uint8_t *x = extra + 768 * (e / 36);
        const int k = e % 36;
        if (k < 32) fs_embed(r, x + 32 * (k >> 1) + 16 * (k & 1), 16);
        else fs_embed(r, x + 512 + 48 * (k - 32), 48);
        return;
    }
    e -= 36 * m1;
    if (e < 14) { fs_embed(r, comms + 48 * (4 + e), 48); return; }
    e -= 14;
    if (e < 21) { fs_embed(r, &p->evals[e], 32); return; }
    e -= 21;
    if (e < 2 * m1) { fs_embed(r, extra + 768 * (e >> 1) + 704 + 32 * (e & 1), 32); return; }
    e -= 2 * m1;
    fs_embed(r, &pt->parts[e / ORC_VS_VK_COMMS].vk_evals[e % ORC_VS_VK_COMMS], 32);
}
static void vs_challenges_poseidon(const orc_vs_circuit_vk *cvk, const orc_vs_proof *p, const uint8_t *extra, int m1, const vs_parts *pt,
                                   fr_t *z_beta, fr_t *z_gamma, fr_t *xi) {
    orc_fq st[3];
    memset(st, 0, sizeof st);
    const int d1 = pt->d1;
    const int n_el = 73 + 38 * m1 + 14 * d1, n_blocks = 37 + 19 * m1 + 7 * d1, b_beta = 21 + 18 * m1 + d1, b_gamma = 25 + 18 * m1 + d1;
    for (int b = 0; b < n_blocks; b++) {
        fq_t e;
        fs_element(&e, cvk, p, extra, m1, pt, 2 * b);
        fq_add((fq_t *)&st[1], (fq_t *)&st[1], &e);
        if (2 * b + 1 < n_el) {
            fs_element(&e, cvk, p, extra, m1, pt, 2 * b + 1);
            fq_add((fq_t *)&st[2], (fq_t *)&st[2], &e);
        }
        orc_poseidon_permute(st);
        if (b == b_beta || b == b_gamma || b == n_blocks - 1) {
            uint64_t c[6];
            fq_to_canonical(c, (const fq_t *)&st[1]);
            fr_from_u128le(b == b_beta ? z_beta : (b == b_gamma ? z_gamma : xi), (const uint8_t *)c); /* low 128 bits */
        }
    }
}

/* what snarkVM's deserialiser guarantees and a raw-coordinate ABI must check itself (include/mi355x_snark.h,
 * "PRECONDITIONS ON POINTS"): coordinates < q, points on the curve or (0,0), Fr fields < r */
static int limbs_lt(const uint64_t *a, const uint64_t *m, int n) {
    for (int i = n - 1; i >= 0; i--) {
        if (a[i] < m[i]) return 1;
        if (a[i] > m[i]) return 0;
    }
    return 0;
}
int orc_vs_proof_is_well_formed(const orc_vs_proof *p) {
    for (int k = 0; k < ORC_VS_INPUTS; k++)
        if (!limbs_lt((const uint64_t *)&p->inputs[k], ORC_FR_MODULUS, 4)) return 0;
    for (int k = 0; k < ORC_VS_EVALS; k++)
        if (!limbs_lt((const uint64_t *)&p->evals[k], ORC_FR_MODULUS, 4)) return 0;
    for (int k = 0; k < ORC_VS_PROOF_COMMS + 2; k++) {
        const orc_g1_affine *pt = k < ORC_VS_PROOF_COMMS ? &p->comms[k] : &p->w[k - ORC_VS_PROOF_COMMS];
        if (!limbs_lt((const uint64_t *)&pt->x, ORC_FQ_MODULUS, 6) || !limbs_lt((const uint64_t *)&pt->y, ORC_FQ_MODULUS, 6)) return 0;
        if (!orc_g1_is_on_curve(pt)) return 0;
    }
    return 1;
}

/* the transcript with m1 further instances (extra: m1 x 768 bytes): each is chained in after C_1 (inputs and
 * commitments) and after the evaluations (its two) */
static void vs_challenges_pt(const orc_vs_circuit_vk *cvk, const orc_vs_proof *p, const uint8_t *extra, int m1, const vs_parts *pt,
                             orc_fr *z_beta, orc_fr *z_gamma, orc_fr *xi);
static void vs_challenges_ex(const orc_vs_circuit_vk *cvk, const orc_vs_proof *p, const uint8_t *extra, int m1, orc_fr *z_beta,
                             orc_fr *z_gamma, orc_fr *xi) {
    vs_challenges_pt(cvk, p, extra, m1, &VS_NO_PARTS, z_beta, z_gamma, xi);
}
void orc_vs_challenges(const orc_vs_circuit_vk *cvk, const orc_vs_proof *p, orc_fr *z_beta, orc_fr *z_gamma,
                       orc_fr *xi) {
    vs_challenges_ex(cvk, p, NULL, 0, z_beta, z_gamma, xi);
}
static void vs_challenges_pt(const orc_vs_circuit_vk *cvk, const orc_vs_proof *p, const uint8_t *extra, int m1, const vs_parts *pt,
                             orc_fr *z_beta, orc_fr *z_gamma, orc_fr *xi) {
    if (g_vs_fs_kind) {
        vs_challenges_poseidon(cvk, p, extra, m1, pt, S(z_beta), S(z_gamma), S(xi));
        return;
    }
    uint8_t buf[32 + 21 * 32 + 64], h[32];
    size_t o = 0;
    memcpy(buf + o, VS_TAG, 32); o += 32;
    memcpy(buf + o, cvk->digest, 32); o += 32;
    memcpy(buf + o, p->inputs, sizeof p->inputs); o += sizeof p->inputs;
    memcpy(buf + o, &p->comms[0], 2 * 96); o += 2 * 96;
    orc_sha256(h, buf, o);
    for (int d = 0; d < pt->d1; d++) { /* every further circuit: its header (id, first instance) and digest */
        memcpy(buf, h, 32);
        memcpy(buf + 32, &pt->parts[d], 16);
        memcpy(buf + 48, pt->circuits[pt->parts[d].circuit].digest, 32);
        orc_sha256(h, buf, 80);
    }
    for (int j = 0; j < m1; j++) {
        memcpy(buf, h, 32);
        memcpy(buf + 32, extra + 768 * j, 704);
        orc_sha256(h, buf, 32 + 704);
    }
    o = 0;
    memcpy(buf, h, 32); o = 32;
    memcpy(buf + o, &p->comms[2], 3 * 96); o += 3 * 96;
    orc_sha256(h, buf, o);
    fr_from_u128le(S(z_beta), h);
    memcpy(buf, h, 32); o = 32;
    memcpy(buf + o, &p->comms[5], 4 * 96); o += 4 * 96;
    orc_sha256(h, buf, o);
    fr_from_u128le(S(z_gamma), h);
    memcpy(buf, h, 32); o = 32;
    memcpy(buf + o, p->evals, sizeof p->evals); o += sizeof p->evals;
    orc_sha256(h, buf, o);
    for (int j = 0; j < m1; j++) {
        memcpy(buf, h, 32);
        memcpy(buf + 32, extra + 768 * j + 704, 64);
        orc_sha256(h, buf, 32 + 64);
    }
    for (int d = 0; d < pt->d1; d++) { /* and the evaluations of every further circuit's polynomials */
        memcpy(buf, h, 32);
        memcpy(buf + 32, pt->parts[d].vk_evals, 384);
        orc_sha256(h, buf, 32 + 384);
    }
    fr_from_u128le(S(xi), h);
}

/* x_hat(z) = sum_k inputs[k] L_k(z) over the size-16 domain; returns 0 if z^16 == 1 (proof rejected) */
static int vs_input_eval(fr_t *xhat, fr_t *zn_minus_1, const fr_t *inputs, const fr_t *z) {
    fr_t w, wk[ORC_VS_INPUTS], den[ORC_VS_INPUTS], pre[ORC_VS_INPUTS], acc, one, zn, t, ninv;
    orc_fr_omega((orc_fr *)&w, 4);
    fr_one(&one);
    zn = *z;
    for (int i = 0; i < 4; i++) fr_sqr(&zn, &zn);
    fr_sub(zn_minus_1, &zn, &one);
    if (fr_is_zero(zn_minus_1)) return 0;
    wk[0] = one;
    for (int k = 1; k < ORC_VS_INPUTS; k++) fr_mul(&wk[k], &wk[k - 1], &w);
    acc = one;
    for (int k = 0; k < ORC_VS_INPUTS; k++) {
        fr_sub(&den[k], z, &wk[k]);
        pre[k] = acc;
        fr_mul(&acc, &acc, &den[k]);
    }
    fr_inv(&acc, &acc);
    fr_zero(xhat);
    for (int k = ORC_VS_INPUTS - 1; k >= 0; k--) {
        fr_t dinv;
        fr_mul(&dinv, &acc, &pre[k]);
        fr_mul(&acc, &acc, &den[k]);
        fr_mul(&t, &inputs[k], &wk[k]);
        fr_mul(&t, &t, &dinv);
        fr_add(xhat, xhat, &t);
    }
    fr_from_u64(&ninv, ORC_VS_INPUTS);
    fr_inv(&ninv, &ninv);
    fr_mul(xhat, xhat, zn_minus_1);
    fr_mul(xhat, xhat, &ninv);
    return 1;
}

/* which query (0 = beta, 1 = gamma) polynomial p (0..20: 9 proof comms then 12 vk comms) opens at */
static inline int vs_query_of(int p) {
    if (p < ORC_VS_PROOF_COMMS) return p < ORC_VS_BETA_PROOF_COMMS ? 0 : 1;
    return (p - ORC_VS_PROOF_COMMS) < ORC_VS_BETA_VK_COMMS ? 0 : 1;
}

/* Everything the verifier derives in Fr for one proof: per-polynomial MSM scalar (already times r_q),
 * witness scalars, and the G coefficient.  Returns the algebraic-check verdict. */
typedef struct {
    fr_t poly_scalar[ORC_VS_EVALS]; /* r_q * xi^p */
    fr_t w_scalar_c[2];             /* r_q * z_q     (goes with H)     */
    fr_t w_scalar_w[2];             /* r_q           (goes with tau H) */
    fr_t g_scalar;                  /* -(sum_q r_q val_q) */
    int alg_ok;
} vs_fr_work;

static void vs_derive(vs_fr_work *o, const orc_vs_circuit_vk *cvk, const orc_vs_proof *p, const uint8_t *rand2x16) {
    fr_t z[2], xi, r[2], xip, val[2], t, xhat, znm1;
    orc_vs_challenges(cvk, p, (orc_fr *)&z[0], (orc_fr *)&z[1], (orc_fr *)&xi);
    fr_from_u128le(&r[0], rand2x16);
    fr_from_u128le(&r[1], rand2x16 + 16);
    fr_one(&xip);
    fr_zero(&val[0]);
    fr_zero(&val[1]);
    for (int k = 0; k < ORC_VS_EVALS; k++) {
        int q = vs_query_of(k);
        fr_mul(&o->poly_scalar[k], &r[q], &xip);
        fr_mul(&t, &xip, S(&p->evals[k]));
        fr_add(&val[q], &val[q], &t);
        fr_mul(&xip, &xip, &xi);
    }
    fr_zero(&o->g_scalar);
    for (int q = 0; q < 2; q++) {
        fr_mul(&o->w_scalar_c[q], &r[q], &z[q]);
        o->w_scalar_w[q] = r[q];
        fr_mul(&t, &r[q], &val[q]);
        fr_sub(&o->g_scalar, &o->g_scalar, &t);
    }
    /* public-input consistency: v_0 == x_hat(z_beta) + (z_beta^16 - 1) v_1; a malformed proof fails here too */
    o->alg_ok = 0;
    if (orc_vs_proof_is_well_formed(p) && vs_input_eval(&xhat, &znm1, (const fr_t *)p->inputs, &z[0])) {
        fr_mul(&t, &znm1, S(&p->evals[1]));
        fr_add(&t, &t, &xhat);
        o->alg_ok = fr_eq(&t, S(&p->evals[0]));
    }
}

int orc_vs_verify_one(const orc_kzg_vk *vk, const orc_g2_prepared *h, const orc_g2_prepared *tau_h,
                      const orc_vs_circuit_vk *circuits, const orc_vs_proof *p, const uint8_t *rand2x16) {
    const orc_vs_circuit_vk *cvk = &circuits[p->circuit];
    vs_fr_work fw;
    vs_derive(&fw, cvk, p, rand2x16);
    orc_g1_jac total_c, total_w, t;
    g1_set_inf(&total_c);
    g1_set_inf(&total_w);
    for (int k = 0; k < ORC_VS_EVALS; k++) {
        const orc_g1_affine *c = k < ORC_VS_PROOF_COMMS ? &p->comms[k] : &cvk->comms[k - ORC_VS_PROOF_COMMS];
        g1_mul_fr(&t, c, &fw.poly_scalar[k]);
        orc_g1_add(&total_c, &total_c, &t);
    }
    for (int q = 0; q < 2; q++) {
        g1_mul_fr(&t, &p->w[q], &fw.w_scalar_c[q]);
        orc_g1_add(&total_c, &total_c, &t);
        g1_mul_fr(&t, &p->w[q], &fw.w_scalar_w[q]);
        orc_g1_add(&total_w, &total_w, &t);
    }
    g1_mul_fr(&t, &vk->g, &fw.g_scalar);
    orc_g1_add(&total_c, &total_c, &t);
    orc_g1_affine ps[2];
    orc_g1_to_affine(&ps[0], &total_c);
    orc_g1_to_affine(&ps[1], &total_w);
    orc_g1_neg_affine(&ps[1], &ps[1]);
    const orc_g2_prepared *qs[2] = {h, tau_h};
    int kzg_ok = orc_pairing_product_is_one(ps, qs, 2);
    return fw.alg_ok && kzg_ok == 1;
}

void orc_vs_verify_batch_cpu(const orc_kzg_vk *vk, const orc_vs_circuit_vk *circuits, const orc_vs_proof *proofs,
                             size_t n, const uint8_t *randomizers128, uint8_t *verdicts, int threads) {
    orc_g2_prepared *h = (orc_g2_prepared *)malloc(sizeof *h), *th = (orc_g2_prepared *)malloc(sizeof *th);
    orc_g2_prepare(h, &vk->h);
    orc_g2_prepare(th, &vk->tau_h);
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads)
    for (size_t i = 0; i < n; i++)
        verdicts[i] = (uint8_t)orc_vs_verify_one(vk, h, th, circuits, &proofs[i], randomizers128 + 32 * i);
    free(h);
    free(th);
}

int orc_vs_verify_aggregated(const orc_kzg_vk *vk, const orc_vs_circuit_vk *circuits, size_t n_circuits,
                             const orc_vs_proof *proofs, size_t n, const uint8_t *randomizers128,
                             uint8_t *alg_ok, orc_g1_affine *total_c_out, orc_g1_affine *total_w_out, int threads) {
    /* one random-linear-combination MSM over all proofs, vk commitments merged per circuit, G merged */
    size_t per = ORC_VS_PROOF_COMMS + 2;
    size_t npts = n * per + n_circuits * ORC_VS_VK_COMMS + 1;
    orc_g1_affine *pts = (orc_g1_affine *)malloc(sizeof(orc_g1_affine) * npts);
    orc_bigint256 *sc = (orc_bigint256 *)malloc(sizeof(orc_bigint256) * npts);
    orc_g1_affine *wpts = (orc_g1_affine *)malloc(sizeof(orc_g1_affine) * n * 2);
    orc_bigint256 *wsc = (orc_bigint256 *)malloc(sizeof(orc_bigint256) * n * 2);
    fr_t *vk_acc = (fr_t *)calloc(n_circuits * ORC_VS_VK_COMMS, sizeof(fr_t));
    fr_t g_acc;
    fr_zero(&g_acc);
    int all_alg = 1;
    for (size_t i = 0; i < n; i++) {
        const orc_vs_proof *p = &proofs[i];
        vs_fr_work fw;
        vs_derive(&fw, &circuits[p->circuit], p, randomizers128 + 32 * i);
        alg_ok[i] = (uint8_t)fw.alg_ok;
        all_alg &= fw.alg_ok;
        for (int k = 0; k < ORC_VS_PROOF_COMMS; k++) {
            pts[i * per + k] = p->comms[k];
            fr_to_canonical(sc[i * per + k].l, &fw.poly_scalar[k]);
        }
        for (int q = 0; q < 2; q++) {
            pts[i * per + ORC_VS_PROOF_COMMS + q] = p->w[q];
            fr_to_canonical(sc[i * per + ORC_VS_PROOF_COMMS + q].l, &fw.w_scalar_c[q]);
            wpts[2 * i + q] = p->w[q];
            fr_to_canonical(wsc[2 * i + q].l, &fw.w_scalar_w[q]);
        }
        for (int k = 0; k < ORC_VS_VK_COMMS; k++) {
            fr_t *a = &vk_acc[p->circuit * ORC_VS_VK_COMMS + k];
            fr_add(a, a, &fw.poly_scalar[ORC_VS_PROOF_COMMS + k]);
        }
        fr_add(&g_acc, &g_acc, &fw.g_scalar);
    }
    size_t o = n * per;
    for (size_t c = 0; c < n_circuits; c++)
        for (int k = 0; k < ORC_VS_VK_COMMS; k++, o++) {
            pts[o] = circuits[c].comms[k];
            fr_to_canonical(sc[o].l, &vk_acc[c * ORC_VS_VK_COMMS + k]);
        }
    pts[o] = vk->g;
    fr_to_canonical(sc[o].l, &g_acc);
    orc_g1_jac tc, tw;
    orc_msm_pippenger(&tc, pts, sc, npts, threads);
    orc_msm_pippenger(&tw, wpts, wsc, n * 2, threads);
    orc_g1_affine ps[2];
    orc_g1_to_affine(&ps[0], &tc);
    orc_g1_to_affine(&ps[1], &tw);
    if (total_c_out) *total_c_out = ps[0];
    if (total_w_out) *total_w_out = ps[1];
    orc_g1_neg_affine(&ps[1], &ps[1]);
    orc_g2_prepared *h = (orc_g2_prepared *)malloc(sizeof *h), *th = (orc_g2_prepared *)malloc(sizeof *th);
    orc_g2_prepare(h, &vk->h);
    orc_g2_prepare(th, &vk->tau_h);
    const orc_g2_prepared *qs[2] = {h, th};
    int ok = orc_pairing_product_is_one(ps, qs, 2);
    free(h); free(th); free(pts); free(sc); free(wpts); free(wsc); free(vk_acc);
    return ok == 1 && all_alg;
}

/* ---- synthetic prover (toy SRS: tau known) ---- */
typedef struct { uint64_t s; } smix;
static uint64_t smix_next(smix *g) {
    g->s += 0x9E3779B97F4A7C15ull;
    uint64_t z = g->s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static void smix_fr(smix *g, fr_t *out) { /* 3 words + a 60-bit top word: always < r, then to Montgomery */
    uint64_t l[4];
    for (int i = 0; i < 4; i++) l[i] = smix_next(g);
    l[3] &= 0x0fffffffffffffffull;
    fr_from_canonical(out, l);
}
#define VS_DEG 4 /* coefficients per synthetic polynomial */
static void poly_eval(fr_t *r, const fr_t *c, const fr_t *x) {
    *r = c[VS_DEG - 1];
    for (int i = VS_DEG - 2; i >= 0; i--) { fr_mul(r, r, x); fr_add(r, r, &c[i]); }
}

void orc_vs_generate(uint64_t seed, size_t n, size_t n_circuits, uint32_t corrupt_per_1024, orc_kzg_vk *vk,
                     orc_vs_circuit_vk *circuits, orc_vs_proof *proofs, uint8_t *randomizers128, uint8_t *expected) {
    smix g = {seed};
    fr_t tau;
    smix_fr(&g, &tau);
    orc_g1_generator(&vk->g);
    orc_g2_generator(&vk->h);
    orc_bigint256 tau_c;
    fr_to_canonical(tau_c.l, &tau);
    orc_g2_mul(&vk->tau_h, &vk->h, &tau_c);

    /* circuit polynomials and commitments */
    fr_t *cpoly = (fr_t *)malloc(sizeof(fr_t) * n_circuits * ORC_VS_VK_COMMS * VS_DEG);
    fr_t *ctau = (fr_t *)malloc(sizeof(fr_t) * n_circuits * ORC_VS_VK_COMMS);
    for (size_t i = 0; i < n_circuits * ORC_VS_VK_COMMS; i++) {
        for (int d = 0; d < VS_DEG; d++) smix_fr(&g, &cpoly[i * VS_DEG + d]);
        poly_eval(&ctau[i], &cpoly[i * VS_DEG], &tau);
    }
    orc_g1_affine *ccomm = (orc_g1_affine *)malloc(sizeof(orc_g1_affine) * n_circuits * ORC_VS_VK_COMMS);
    orc_g1_fixed_base_batch(ccomm, (const orc_fr *)ctau, n_circuits * ORC_VS_VK_COMMS);
    for (size_t c = 0; c < n_circuits; c++) {
        memcpy(circuits[c].comms, &ccomm[c * ORC_VS_VK_COMMS], sizeof circuits[c].comms);
        orc_sha256(circuits[c].digest, (const uint8_t *)circuits[c].comms, sizeof circuits[c].comms);
    }

    /* proof polynomials: p_1..p_8 random; p_0 = x_hat + (X^16-1) p_1 handled through its values */
    fr_t *ppoly = (fr_t *)malloc(sizeof(fr_t) * n * ORC_VS_PROOF_COMMS * VS_DEG);
    fr_t *ptau = (fr_t *)malloc(sizeof(fr_t) * n * ORC_VS_PROOF_COMMS);
    for (size_t i = 0; i < n; i++) {
        orc_vs_proof *p = &proofs[i];
        memset(p, 0, sizeof *p);
        p->circuit = (uint32_t)(smix_next(&g) % n_circuits);
        for (int k = 0; k < ORC_VS_INPUTS; k++) smix_fr(&g, S(&p->inputs[k]));
        for (int k = 0; k < ORC_VS_PROOF_COMMS; k++)
            for (int d = 0; d < VS_DEG; d++) smix_fr(&g, &ppoly[(i * ORC_VS_PROOF_COMMS + k) * VS_DEG + d]);
        for (int k = 1; k < ORC_VS_PROOF_COMMS; k++)
            poly_eval(&ptau[i * ORC_VS_PROOF_COMMS + k], &ppoly[(i * ORC_VS_PROOF_COMMS + k) * VS_DEG], &tau);
        fr_t xh, znm1, t;
        vs_input_eval(&xh, &znm1, (const fr_t *)p->inputs, &tau);
        fr_mul(&t, &znm1, &ptau[i * ORC_VS_PROOF_COMMS + 1]);
        fr_add(&ptau[i * ORC_VS_PROOF_COMMS + 0], &xh, &t);
        for (int k = 0; k < 32; k++) randomizers128[32 * i + k] = (uint8_t)smix_next(&g);
    }
    orc_g1_affine *pcomm = (orc_g1_affine *)malloc(sizeof(orc_g1_affine) * n * ORC_VS_PROOF_COMMS);
    orc_g1_fixed_base_batch(pcomm, (const orc_fr *)ptau, n * ORC_VS_PROOF_COMMS);
    for (size_t i = 0; i < n; i++) memcpy(proofs[i].comms, &pcomm[i * ORC_VS_PROOF_COMMS], sizeof proofs[i].comms);

    /* evaluations: z_beta, z_gamma depend only on inputs + commitments */
    fr_t *wk = (fr_t *)malloc(sizeof(fr_t) * n * 2);
    for (size_t i = 0; i < n; i++) {
        orc_vs_proof *p = &proofs[i];
        const orc_vs_circuit_vk *cvk = &circuits[p->circuit];
        fr_t z[2], xi, xip, t, num[2];
        orc_vs_challenges(cvk, p, (orc_fr *)&z[0], (orc_fr *)&z[1], (orc_fr *)&xi);
        for (int k = 0; k < ORC_VS_EVALS; k++) {
            int q = vs_query_of(k);
            const fr_t *poly = k < ORC_VS_PROOF_COMMS
                                   ? &ppoly[(i * ORC_VS_PROOF_COMMS + k) * VS_DEG]
                                   : &cpoly[(p->circuit * ORC_VS_VK_COMMS + (k - ORC_VS_PROOF_COMMS)) * VS_DEG];
            poly_eval(S(&p->evals[k]), poly, &z[q]);
        }
        { /* p_0(z_beta) = x_hat(z_beta) + (z_beta^16-1) p_1(z_beta) */
            fr_t xh, znm1;
            if (vs_input_eval(&xh, &znm1, (const fr_t *)p->inputs, &z[0])) {
                fr_mul(&t, &znm1, S(&p->evals[1]));
                fr_add(S(&p->evals[0]), &xh, &t);
            }
        }
        /* xi depends on the evaluations: recompute, then the witnesses */
        orc_vs_challenges(cvk, p, (orc_fr *)&z[0], (orc_fr *)&z[1], (orc_fr *)&xi);
        fr_one(&xip);
        fr_zero(&num[0]);
        fr_zero(&num[1]);
        for (int k = 0; k < ORC_VS_EVALS; k++) {
            int q = vs_query_of(k);
            const fr_t *at_tau = k < ORC_VS_PROOF_COMMS ? &ptau[i * ORC_VS_PROOF_COMMS + k]
                                                          : &ctau[p->circuit * ORC_VS_VK_COMMS + (k - ORC_VS_PROOF_COMMS)];
            fr_sub(&t, at_tau, S(&p->evals[k]));
            fr_mul(&t, &t, &xip);
            fr_add(&num[q], &num[q], &t);
            fr_mul(&xip, &xip, &xi);
        }
        for (int q = 0; q < 2; q++) {
            fr_sub(&t, &tau, &z[q]);
            fr_inv(&t, &t);
            fr_mul(&wk[2 * i + q], &num[q], &t);
        }
    }
    orc_g1_affine *wc = (orc_g1_affine *)malloc(sizeof(orc_g1_affine) * n * 2);
    orc_g1_fixed_base_batch(wc, (const orc_fr *)wk, n * 2);
    for (size_t i = 0; i < n; i++) {
        proofs[i].w[0] = wc[2 * i];
        proofs[i].w[1] = wc[2 * i + 1];
        expected[i] = 1;
    }
    /* corruption: evaluations changed AFTER the witnesses were made (xi changes too), a witness swapped,
     * or a public input changed */
    for (size_t i = 0; i < n; i++) {
        if ((smix_next(&g) & 1023) >= corrupt_per_1024) continue;
        orc_vs_proof *p = &proofs[i];
        fr_t one;
        fr_one(&one);
        switch (smix_next(&g) % 4) {
        case 0: { int k = 2 + (int)(smix_next(&g) % (ORC_VS_EVALS - 2)); fr_add(S(&p->evals[k]), S(&p->evals[k]), &one); break; }
        case 1: p->w[smix_next(&g) & 1] = vk->g; break;
        case 2: fr_add(S(&p->inputs[smix_next(&g) % ORC_VS_INPUTS]), S(&p->inputs[0]), &one); break;
        default: p->comms[2 + smix_next(&g) % 7] = vk->g; break;
        }
        expected[i] = 0;
    }
    free(cpoly); free(ctau); free(ccomm); free(ppoly); free(ptau); free(pcomm); free(wk); free(wc);
}

/* ------------------------------------------------------------------ VarunaShaped-v2
 * The shape of SonicKZG10::check_combinations on top of v1:
 *   - the query set is a TABLE (shape->query_of[p]: which of the two query points each of the 21 polynomials opens at),
 *     not the hard-wired split of v1;
 *   - hiding: commitments carry rho_p gamma_g, the proof carries random_v[q] = sum_{p at q} xi^p rho_p, the check
 *     subtracts (sum_q r_q random_v[q]) gamma_g (kzg10::VerifierKey.gamma_g, KZGProof.random_v);
 *   - degree bounds: proof polynomial j of class c = shape->bound_class[j] comes with a shifted commitment S_j; the
 *     check e(C_j, shift_h[c]) = e(S_j, h) is folded with rho_j = r_3 xi^(21 + j): the product has 2 + n_bounds pairs
 *     (total_c - sum rho_j S_j, h), (-total_w, tau h), (sum_{j in c} rho_j C_j, shift_h[c]).
 * Three 128-bit caller randomizers per proof (r_beta, r_gamma, r_3).  The transcript (either kind) runs over the v1 part
 * of the record.  Toy SRS as in v1: tau, gamma and the shift factors s_c (standing for tau^(D - d_c)) are known to the
 * generator only. */
/* kap[kind] = (z_beta^16 - 1)^(kind & 1) (z_gamma^16 - 1)^(kind >> 1): the challenge-dependent factors of the shape's
 * linear-combination table */
/* kinds 4..7 (VarunaShaped-v3): the same times 1 / (z_beta - z_gamma).  Returns 0 when the shape uses such a kind and
 * z_beta == z_gamma (the proof is invalid; kap[4..7] are then zero). */
static int vs2_shape_needs_inverse(const orc_vs_shape *sh) {
    for (int k = 0; k < ORC_VS_EVALS; k++)
        if (sh->lc_kind[k] & 4) return 1;
    return 0;
}
static int vs2_lc_factors(fr_t kap[8], const fr_t z[2], int need_inverse) {
    fr_t v[2], one, d;
    fr_one(&one);
    for (int q = 0; q < 2; q++) {
        v[q] = z[q];
        for (int k = 0; k < 4; k++) fr_mul(&v[q], &v[q], &v[q]);
        fr_sub(&v[q], &v[q], &one);
    }
    kap[0] = one;
    kap[1] = v[0];
    kap[2] = v[1];
    fr_mul(&kap[3], &v[0], &v[1]);
    for (int k = 4; k < 8; k++) fr_zero(&kap[k]);
    if (!need_inverse) return 1;
    fr_sub(&d, &z[0], &z[1]);
    if (fr_is_zero(&d)) return 0;
    fr_inv(&d, &d);
    for (int k = 0; k < 4; k++) fr_mul(&kap[4 + k], &kap[k], &d);
    return 1;
}
/* inst / m1: the further instances of a proof with instances (their scalars r_beta xi^(30 + 2j + c) go to ex_scalar) */
/* pt / px_scalar: the further circuits of a VarunaShaped-v3 proof (their scalars r_q xi^(36 + 12 d + k) kap go to px_scalar) */
static void vs2_derive(vs_fr_work *o, fr_t rho[ORC_VS_PROOF_COMMS], fr_t *h_scalar, const orc_vs_shape *sh,
                       const orc_vs_circuit_vk *cvk, const orc_vs_proof2 *p2, const uint8_t *rand3x16,
                       const orc_vs_instance *inst, int m1, fr_t ex_scalar[][2], const vs_parts *pt,
                       fr_t px_scalar[][ORC_VS_VK_COMMS]) {
    const orc_vs_proof *p = &p2->base;
    fr_t z[2], xi, r[3], xip, val[2], t, xhat, znm1;
    vs_challenges_pt(cvk, p, (const uint8_t *)inst, m1, pt, (orc_fr *)&z[0], (orc_fr *)&z[1], (orc_fr *)&xi);
    for (int k = 0; k < 3; k++) fr_from_u128le(&r[k], rand3x16 + 16 * k);
    fr_one(&xip);
    fr_zero(&val[0]);
    fr_zero(&val[1]);
    fr_t kap[8];
    const int inverse_ok = vs2_lc_factors(kap, z, vs2_shape_needs_inverse(sh));
    for (int k = 0; k < ORC_VS_EVALS; k++) {
        const int q = sh->query_of[k] & 1;
        fr_t coef;
        fr_mul(&coef, &xip, &kap[sh->lc_kind[k] & 7]);
        fr_mul(&o->poly_scalar[k], &r[q], &coef);
        fr_mul(&t, &coef, S(&p->evals[k]));
        fr_add(&val[q], &val[q], &t);
        fr_mul(&xip, &xip, &xi);
    }
    for (int j = 0; j < ORC_VS_PROOF_COMMS; j++) { /* xip = xi^(21 + j) */
        fr_mul(&rho[j], &r[2], &xip);
        fr_mul(&xip, &xip, &xi);
    }
    for (int j = 0; j < m1; j++)      /* xip = xi^(30 + 2j + c): the further instances' polynomials, beta query */
        for (int c = 0; c < 2; c++) {
            fr_mul(&ex_scalar[j][c], &r[0], &xip);
            fr_mul(&t, &xip, S(&inst[j].evals[c]));
            fr_add(&val[0], &val[0], &t);
            fr_mul(&xip, &xip, &xi);
        }
    if (pt->d1) { /* the further circuits' polynomials: xi^(36 + 12 d + k), the first circuit's query and lc tables */
        for (int j = m1; j < ORC_VS_MAX_INSTANCES - 1; j++) { fr_mul(&xip, &xip, &xi); fr_mul(&xip, &xip, &xi); } /* to xi^36 */
        for (int d = 0; d < pt->d1; d++)
            for (int k = 0; k < ORC_VS_VK_COMMS; k++) {
                const int pp = ORC_VS_PROOF_COMMS + k, q = sh->query_of[pp] & 1;
                fr_t coef;
                fr_mul(&coef, &xip, &kap[sh->lc_kind[pp] & 7]);
                fr_mul(&px_scalar[d][k], &r[q], &coef);
                fr_mul(&t, &coef, S(&pt->parts[d].vk_evals[k]));
                fr_add(&val[q], &val[q], &t);
                fr_mul(&xip, &xip, &xi);
            }
    }
    fr_zero(&o->g_scalar);
    fr_zero(h_scalar);
    for (int q = 0; q < 2; q++) {
        fr_mul(&o->w_scalar_c[q], &r[q], &z[q]);
        o->w_scalar_w[q] = r[q];
        fr_mul(&t, &r[q], &val[q]);
        fr_sub(&o->g_scalar, &o->g_scalar, &t);
        if (sh->hiding) {
            fr_mul(&t, &r[q], S(&p2->random_v[q]));
            fr_sub(h_scalar, h_scalar, &t);
        }
    }
    o->alg_ok = 0;
    int wf = orc_vs_proof_is_well_formed(p);
    for (int q = 0; q < 2 && wf; q++) wf = limbs_lt((const uint64_t *)&p2->random_v[q], ORC_FR_MODULUS, 4);
    for (int j = 0; j < ORC_VS_PROOF_COMMS && wf; j++) {
        const orc_g1_affine *pt = &p2->shifted[j];
        wf = limbs_lt((const uint64_t *)&pt->x, ORC_FQ_MODULUS, 6) && limbs_lt((const uint64_t *)&pt->y, ORC_FQ_MODULUS, 6) &&
             orc_g1_is_on_curve(pt);
    }
    for (int j = 0; j < m1 && wf; j++) {
        for (int k = 0; k < ORC_VS_INPUTS && wf; k++) wf = limbs_lt((const uint64_t *)&inst[j].inputs[k], ORC_FR_MODULUS, 4);
        for (int k = 0; k < 2 && wf; k++) wf = limbs_lt((const uint64_t *)&inst[j].evals[k], ORC_FR_MODULUS, 4);
        for (int c = 0; c < 2 && wf; c++) {
            const orc_g1_affine *pt = &inst[j].comms[c];
            wf = limbs_lt((const uint64_t *)&pt->x, ORC_FQ_MODULUS, 6) && limbs_lt((const uint64_t *)&pt->y, ORC_FQ_MODULUS, 6) &&
                 orc_g1_is_on_curve(pt);
        }
    }
    for (int d = 0; d < pt->d1 && wf; d++)
        for (int k = 0; k < ORC_VS_VK_COMMS && wf; k++) wf = limbs_lt((const uint64_t *)&pt->parts[d].vk_evals[k], ORC_FR_MODULUS, 4);
    if (wf && inverse_ok && vs_input_eval(&xhat, &znm1, (const fr_t *)p->inputs, &z[0])) {
        fr_mul(&t, &znm1, S(&p->evals[1]));
        fr_add(&t, &t, &xhat);
        o->alg_ok = fr_eq(&t, S(&p->evals[0]));
        for (int j = 0; j < m1 && o->alg_ok; j++) { /* every instance's own public-input identity */
            vs_input_eval(&xhat, &znm1, (const fr_t *)inst[j].inputs, &z[0]);
            fr_mul(&t, &znm1, S(&inst[j].evals[1]));
            fr_add(&t, &t, &xhat);
            o->alg_ok = fr_eq(&t, S(&inst[j].evals[0]));
        }
    }
}
/* the record of proof i under the shape: orc_vs_proof2, or orc_vs_proof3 when the shape carries instances; *m1 = its
 * further instances (0 when the count is out of range: the proof is then rejected by vs3_count_ok) */
static int vs3_kind(const orc_vs_shape *sh) { return !sh ? 1 : (sh->max_circuits > 1 ? 4 : (sh->max_instances > 1 ? 3 : 2)); }
static size_t vs3_stride(const orc_vs_shape *sh) {
    static const size_t bytes[5] = {0, sizeof(orc_vs_proof), sizeof(orc_vs_proof2), sizeof(orc_vs_proof3), sizeof(orc_vs_proof4)};
    return bytes[vs3_kind(sh)];
}
size_t orc_vs_record_bytes(const orc_vs_shape *sh) { return vs3_stride(sh); }
/* pt: the further circuits (VarunaShaped-v3 records); n_ctx_circuits: circuits of the context.  A record whose instance
 * or circuit structure is out of range is rejected (count_ok = 0) and keeps its first instance and circuit only. */
static const orc_vs_proof2 *vs3_record(const orc_vs_shape *sh, const void *records, size_t i, const orc_vs_instance **inst, int *m1,
                                        int *count_ok, vs_parts *pt, const orc_vs_circuit_vk *circuits, size_t n_ctx_circuits) {
    const uint8_t *rec = (const uint8_t *)records + vs3_stride(sh) * i;
    *inst = NULL;
    *m1 = 0;
    *count_ok = 1;
    pt->parts = NULL;
    pt->d1 = 0;
    pt->circuits = circuits;
    if (sh->max_instances > 1 || sh->max_circuits > 1) {
        const orc_vs_proof3 *p3 = (const orc_vs_proof3 *)rec;
        const uint32_t mi = sh->max_instances > 1 ? sh->max_instances : 1;
        *count_ok = p3->n_instances >= 1 && p3->n_instances <= mi && p3->n_instances <= ORC_VS_MAX_INSTANCES;
        if (*count_ok && sh->max_circuits > 1) {
            const orc_vs_proof4 *p4 = (const orc_vs_proof4 *)rec;
            const uint32_t nc = p4->n_circuits;
            int ok = nc >= 1 && nc <= sh->max_circuits && nc <= ORC_VS_MAX_CIRCUITS && nc <= p3->n_instances &&
                     p3->base2.base.circuit < n_ctx_circuits;
            uint32_t c_prev = p3->base2.base.circuit, f_prev = 0;
            for (uint32_t d = 0; ok && d + 1 < nc; d++) {
                const orc_vs_circuit_part *cp = &p4->more[d];
                ok = cp->circuit > c_prev && cp->circuit < n_ctx_circuits && cp->first_instance > f_prev && cp->first_instance < p3->n_instances;
                c_prev = cp->circuit;
                f_prev = cp->first_instance;
            }
            *count_ok = ok;
            if (ok) {
                pt->parts = p4->more;
                pt->d1 = (int)nc - 1;
            }
        }
        if (*count_ok) {
            *m1 = (int)p3->n_instances - 1;
            *inst = p3->extra;
        }
    }
    return (const orc_vs_proof2 *)rec;
}
/* the four accumulators of one proof (Jacobian) */
static void vs2_accumulate_one(orc_g1_jac acc[4], int *alg_ok, const orc_kzg_vk *vk, const orc_vs_key_ext *ext,
                               const orc_vs_shape *sh, const orc_vs_circuit_vk *circuits, const orc_vs_proof2 *p2,
                               const uint8_t *rand3x16, const orc_vs_instance *inst, int m1, const vs_parts *pt) {
    const orc_vs_proof *p = &p2->base;
    const orc_vs_circuit_vk *cvk = &circuits[p->circuit];
    vs_fr_work fw;
    fr_t rho[ORC_VS_PROOF_COMMS], hs, t, exs[ORC_VS_MAX_INSTANCES][2], pxs[ORC_VS_MAX_CIRCUITS][ORC_VS_VK_COMMS];
    vs2_derive(&fw, rho, &hs, sh, cvk, p2, rand3x16, inst, m1, exs, pt, pxs);
    orc_g1_jac tj;
    for (int k = 0; k < 4; k++) g1_set_inf(&acc[k]);
    for (int k = 0; k < ORC_VS_EVALS; k++) {
        const orc_g1_affine *c = k < ORC_VS_PROOF_COMMS ? &p->comms[k] : &cvk->comms[k - ORC_VS_PROOF_COMMS];
        g1_mul_fr(&tj, c, &fw.poly_scalar[k]);
        orc_g1_add(&acc[0], &acc[0], &tj);
    }
    for (int j = 0; j < m1; j++)
        for (int c = 0; c < 2; c++) {
            g1_mul_fr(&tj, &inst[j].comms[c], &exs[j][c]);
            orc_g1_add(&acc[0], &acc[0], &tj);
        }
    for (int d = 0; d < pt->d1; d++)
        for (int k = 0; k < ORC_VS_VK_COMMS; k++) {
            g1_mul_fr(&tj, &circuits[pt->parts[d].circuit].comms[k], &pxs[d][k]);
            orc_g1_add(&acc[0], &acc[0], &tj);
        }
    for (int q = 0; q < 2; q++) {
        g1_mul_fr(&tj, &p->w[q], &fw.w_scalar_c[q]);
        orc_g1_add(&acc[0], &acc[0], &tj);
        g1_mul_fr(&tj, &p->w[q], &fw.w_scalar_w[q]);
        orc_g1_add(&acc[1], &acc[1], &tj);
    }
    g1_mul_fr(&tj, &vk->g, &fw.g_scalar);
    orc_g1_add(&acc[0], &acc[0], &tj);
    if (sh->hiding) {
        g1_mul_fr(&tj, &ext->gamma_g, &hs);
        orc_g1_add(&acc[0], &acc[0], &tj);
    }
    for (int j = 0; j < ORC_VS_PROOF_COMMS; j++) {
        const int c = sh->bound_class[j];
        if (c < 0 || c >= sh->n_bounds) continue;
        fr_neg(&t, &rho[j]);
        g1_mul_fr(&tj, &p2->shifted[j], &t);
        orc_g1_add(&acc[0], &acc[0], &tj);
        g1_mul_fr(&tj, &p->comms[j], &rho[j]);
        orc_g1_add(&acc[2 + c], &acc[2 + c], &tj);
    }
    *alg_ok = fw.alg_ok;
}
static int vs2_product_is_one(const orc_g1_jac acc[4], const orc_g2_prepared *const preps[4], int n_bounds) {
    orc_g1_affine ps[4];
    for (int k = 0; k < 2 + n_bounds; k++) orc_g1_to_affine(&ps[k], &acc[k]);
    orc_g1_neg_affine(&ps[1], &ps[1]); /* infinity is (0, 0) and stays so */
    return orc_pairing_product_is_one(ps, preps, (size_t)(2 + n_bounds)) == 1;
}
void orc_vs3_verify_batch_cpu(const orc_kzg_vk *vk, const orc_vs_key_ext *ext, const orc_vs_shape *sh,
                              const orc_vs_circuit_vk *circuits, const void *records, size_t n,
                              const uint8_t *randomizers /* n x 48 */, uint8_t *verdicts, int threads) {
    orc_vs4_verify_batch_cpu(vk, ext, sh, circuits, (size_t)-1, records, n, randomizers, NULL, verdicts, threads);
}
void orc_vs4_verify_batch_cpu(const orc_kzg_vk *vk, const orc_vs_key_ext *ext, const orc_vs_shape *sh,
                              const orc_vs_circuit_vk *circuits, size_t n_circuits, const void *records, size_t n,
                              const uint8_t *randomizers /* n x 48 */, const uint8_t *dec_ok, uint8_t *verdicts, int threads) {
    orc_g2_prepared *pr = (orc_g2_prepared *)malloc(sizeof(orc_g2_prepared) * 4);
    orc_g2_prepare(&pr[0], &vk->h);
    orc_g2_prepare(&pr[1], &vk->tau_h);
    for (int c = 0; c < sh->n_bounds; c++) orc_g2_prepare(&pr[2 + c], &ext->shift_h[c]);
    const orc_g2_prepared *preps[4] = {&pr[0], &pr[1], &pr[2], &pr[3]};
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads)
    for (size_t i = 0; i < n; i++) {
        orc_g1_jac acc[4];
        int alg, m1, count_ok;
        const orc_vs_instance *inst;
        vs_parts pt;
        const orc_vs_proof2 *p2 = vs3_record(sh, records, i, &inst, &m1, &count_ok, &pt, circuits, n_circuits);
        if (dec_ok && !dec_ok[i]) { verdicts[i] = 0; continue; }
        vs2_accumulate_one(acc, &alg, vk, ext, sh, circuits, p2, randomizers + 48 * i, inst, m1, &pt);
        verdicts[i] = (uint8_t)(alg && count_ok && vs2_product_is_one(acc, preps, sh->n_bounds));
    }
    free(pr);
}
void orc_vs2_verify_batch_cpu(const orc_kzg_vk *vk, const orc_vs_key_ext *ext, const orc_vs_shape *sh,
                              const orc_vs_circuit_vk *circuits, const orc_vs_proof2 *proofs, size_t n,
                              const uint8_t *randomizers, uint8_t *verdicts, int threads) {
    orc_vs3_verify_batch_cpu(vk, ext, sh, circuits, proofs, n, randomizers, verdicts, threads);
}
/* the aggregated form: per-proof accumulators summed (the group result is what the library's RLC MSMs produce);
 * totals[0..3] = total_c, total_w, bound[0], bound[1] as affine points ((0,0) = infinity).  A proof whose instance
 * count is out of range counts as algebraically failed and contributes its first instance only (what the library's
 * derive kernel does: the further instances' scalars are zero). */
int orc_vs3_verify_aggregated(const orc_kzg_vk *vk, const orc_vs_key_ext *ext, const orc_vs_shape *sh,
                              const orc_vs_circuit_vk *circuits, const void *records, size_t n,
                              const uint8_t *randomizers, uint8_t *alg_ok, orc_g1_affine totals[4], int threads) {
    return orc_vs4_verify_aggregated(vk, ext, sh, circuits, (size_t)-1, records, n, randomizers, NULL, alg_ok, totals, threads);
}
int orc_vs4_verify_aggregated(const orc_kzg_vk *vk, const orc_vs_key_ext *ext, const orc_vs_shape *sh,
                              const orc_vs_circuit_vk *circuits, size_t n_circuits, const void *records, size_t n,
                              const uint8_t *randomizers, const uint8_t *dec_ok, uint8_t *alg_ok, orc_g1_affine totals[4],
                              int threads) {
    if (threads < 1) threads = 1;
    orc_g1_jac *part = (orc_g1_jac *)malloc(sizeof(orc_g1_jac) * 4 * (size_t)threads);
    for (int k = 0; k < 4 * threads; k++) g1_set_inf(&part[k]);
#pragma omp parallel num_threads(threads)
    {
        const int tid = omp_get_thread_num();
#pragma omp for schedule(dynamic, 8)
        for (size_t i = 0; i < n; i++) {
            orc_g1_jac acc[4];
            int alg, m1, count_ok;
            const orc_vs_instance *inst;
            vs_parts pt;
            const orc_vs_proof2 *p2 = vs3_record(sh, records, i, &inst, &m1, &count_ok, &pt, circuits, n_circuits);
            vs2_accumulate_one(acc, &alg, vk, ext, sh, circuits, p2, randomizers + 48 * i, inst, m1, &pt);
            alg_ok[i] = (uint8_t)(alg && count_ok && (!dec_ok || dec_ok[i]));
            for (int k = 0; k < 4; k++) orc_g1_add(&part[4 * tid + k], &part[4 * tid + k], &acc[k]);
        }
    }
    orc_g1_jac tot[4];
    for (int k = 0; k < 4; k++) {
        g1_set_inf(&tot[k]);
        for (int t = 0; t < threads; t++) orc_g1_add(&tot[k], &tot[k], &part[4 * t + k]);
    }
    free(part);
    int all_alg = 1;
    for (size_t i = 0; i < n; i++) all_alg &= alg_ok[i];
    orc_g2_prepared *pr = (orc_g2_prepared *)malloc(sizeof(orc_g2_prepared) * 4);
    orc_g2_prepare(&pr[0], &vk->h);
    orc_g2_prepare(&pr[1], &vk->tau_h);
    for (int c = 0; c < sh->n_bounds; c++) orc_g2_prepare(&pr[2 + c], &ext->shift_h[c]);
    const orc_g2_prepared *preps[4] = {&pr[0], &pr[1], &pr[2], &pr[3]};
    const int ok = vs2_product_is_one(tot, preps, sh->n_bounds);
    free(pr);
    for (int k = 0; k < 4; k++) orc_g1_to_affine(&totals[k], &tot[k]); /* (0, 0) for infinity */
    return ok && all_alg;
}
int orc_vs2_verify_aggregated(const orc_kzg_vk *vk, const orc_vs_key_ext *ext, const orc_vs_shape *sh,
                              const orc_vs_circuit_vk *circuits, const orc_vs_proof2 *proofs, size_t n,
                              const uint8_t *randomizers, uint8_t *alg_ok, orc_g1_affine totals[4], int threads) {
    return orc_vs3_verify_aggregated(vk, ext, sh, circuits, proofs, n, randomizers, alg_ok, totals, threads);
}
void orc_vs3_generate(uint64_t seed, size_t n, size_t n_circuits, uint32_t corrupt_per_1024, const orc_vs_shape *sh,
                      orc_kzg_vk *vk, orc_vs_key_ext *ext, orc_vs_circuit_vk *circuits, void *records,
                      uint8_t *randomizers /* n x 48 */, uint8_t *expected) {
    orc_vs4_generate(seed, n, n_circuits, corrupt_per_1024, sh, vk, ext, circuits, records, randomizers, expected);
}
/* Shapes without several circuits draw exactly the random stream they always drew (committed fixtures stay reproducible);
 * with max_circuits > 1 a proof also draws its number of circuits, their ids and first instances (n_circuits of the
 * context must be at least max_circuits), every further circuit's 12 evaluations are made like the first circuit's,
 * and a corrupted proof may have a part spoilt. */
void orc_vs4_generate(uint64_t seed, size_t n, size_t n_circuits, uint32_t corrupt_per_1024, const orc_vs_shape *sh,
                      orc_kzg_vk *vk, orc_vs_key_ext *ext, orc_vs_circuit_vk *circuits, void *records,
                      uint8_t *randomizers /* n x 48 */, uint8_t *expected) {
    const size_t stride = vs3_stride(sh);
    const int MC = sh->max_circuits > 1 ? (sh->max_circuits > ORC_VS_MAX_CIRCUITS ? ORC_VS_MAX_CIRCUITS : sh->max_circuits) : 1;
    const int MI = sh->max_instances > 1 ? (sh->max_instances > ORC_VS_MAX_INSTANCES ? ORC_VS_MAX_INSTANCES : sh->max_instances) : 1;
    const int need_inv = vs2_shape_needs_inverse(sh);
#define REC2(i) ((orc_vs_proof2 *)((uint8_t *)records + stride * (i)))
#define REC3(i) ((orc_vs_proof3 *)((uint8_t *)records + stride * (i)))
#define REC4(i) ((orc_vs_proof4 *)((uint8_t *)records + stride * (i)))
    const size_t EX = (size_t)(ORC_VS_MAX_INSTANCES - 1) * 2; /* further-instance polynomials per proof */
    smix g = {seed};
    fr_t tau, gam, shf[ORC_VS_MAX_BOUNDS];
    smix_fr(&g, &tau);
    smix_fr(&g, &gam);
    for (int c = 0; c < ORC_VS_MAX_BOUNDS; c++) smix_fr(&g, &shf[c]);
    orc_g1_generator(&vk->g);
    orc_g2_generator(&vk->h);
    orc_bigint256 kc;
    fr_to_canonical(kc.l, &tau);
    orc_g2_mul(&vk->tau_h, &vk->h, &kc);
    {
        orc_g1_jac tj;
        g1_mul_fr(&tj, &vk->g, &gam);
        orc_g1_to_affine(&ext->gamma_g, &tj);
    }
    for (int c = 0; c < ORC_VS_MAX_BOUNDS; c++) {
        fr_to_canonical(kc.l, &shf[c]);
        orc_g2_mul(&ext->shift_h[c], &vk->h, &kc);
    }
    /* circuit polynomials and commitments (never hiding, never bounded) */
    fr_t *cpoly = (fr_t *)malloc(sizeof(fr_t) * n_circuits * ORC_VS_VK_COMMS * VS_DEG);
    fr_t *ctau = (fr_t *)malloc(sizeof(fr_t) * n_circuits * ORC_VS_VK_COMMS);
    for (size_t i = 0; i < n_circuits * ORC_VS_VK_COMMS; i++) {
        for (int d = 0; d < VS_DEG; d++) smix_fr(&g, &cpoly[i * VS_DEG + d]);
        poly_eval(&ctau[i], &cpoly[i * VS_DEG], &tau);
    }
    orc_g1_affine *ccomm = (orc_g1_affine *)malloc(sizeof(orc_g1_affine) * n_circuits * ORC_VS_VK_COMMS);
    orc_g1_fixed_base_batch(ccomm, (const orc_fr *)ctau, n_circuits * ORC_VS_VK_COMMS);
    for (size_t c = 0; c < n_circuits; c++) {
        memcpy(circuits[c].comms, &ccomm[c * ORC_VS_VK_COMMS], sizeof circuits[c].comms);
        orc_sha256(circuits[c].digest, (const uint8_t *)circuits[c].comms, sizeof circuits[c].comms);
    }
    const size_t PC = ORC_VS_PROOF_COMMS;
    fr_t *ppoly = (fr_t *)malloc(sizeof(fr_t) * n * PC * VS_DEG);
    fr_t *ptau = (fr_t *)malloc(sizeof(fr_t) * n * PC);   /* p_k(tau) */
    fr_t *prho = (fr_t *)malloc(sizeof(fr_t) * n * PC);   /* hiding constants rho_k */
    fr_t *cval = (fr_t *)malloc(sizeof(fr_t) * n * PC * 2); /* commitment exponents: C_k, then S_k */
    /* further instances (MI > 1): polynomials A_j, B_j, their values at tau, hiding constants, commitment exponents */
    fr_t *xpoly = (fr_t *)calloc(n * EX * VS_DEG + 1, sizeof(fr_t));
    fr_t *xtau = (fr_t *)calloc(n * EX + 1, sizeof(fr_t));
    fr_t *xrho = (fr_t *)calloc(n * EX + 1, sizeof(fr_t));
    fr_t *xcval = (fr_t *)calloc(n * EX + 1, sizeof(fr_t));
    for (size_t i = 0; i < n; i++) {
        orc_vs_proof2 *p2 = REC2(i);
        orc_vs_proof *p = &p2->base;
        memset(p2, 0, stride);
        p->circuit = (uint32_t)(smix_next(&g) % n_circuits);
        for (int k = 0; k < ORC_VS_INPUTS; k++) smix_fr(&g, S(&p->inputs[k]));
        for (size_t k = 0; k < PC; k++)
            for (int d = 0; d < VS_DEG; d++) smix_fr(&g, &ppoly[(i * PC + k) * VS_DEG + d]);
        for (size_t k = 1; k < PC; k++) poly_eval(&ptau[i * PC + k], &ppoly[(i * PC + k) * VS_DEG], &tau);
        fr_t xh, znm1, t;
        vs_input_eval(&xh, &znm1, (const fr_t *)p->inputs, &tau);
        fr_mul(&t, &znm1, &ptau[i * PC + 1]);
        fr_add(&ptau[i * PC + 0], &xh, &t);
        for (size_t k = 0; k < PC; k++) {
            if (sh->hiding) smix_fr(&g, &prho[i * PC + k]);
            else fr_zero(&prho[i * PC + k]);
            fr_mul(&t, &gam, &prho[i * PC + k]);
            fr_add(&cval[(i * PC + k) * 2], &ptau[i * PC + k], &t);
            const int c = sh->bound_class[k];
            if (c >= 0 && c < sh->n_bounds) fr_mul(&cval[(i * PC + k) * 2 + 1], &cval[(i * PC + k) * 2], &shf[c]);
            else fr_zero(&cval[(i * PC + k) * 2 + 1]);
        }
        for (int k = 0; k < 48; k++) randomizers[48 * i + k] = (uint8_t)smix_next(&g);
        if (MI > 1) {
            orc_vs_proof3 *p3 = REC3(i);
            p3->n_instances = 1 + (uint32_t)(smix_next(&g) % (uint64_t)MI);
            for (uint32_t j = 0; j + 1 < p3->n_instances; j++) {
                orc_vs_instance *in = &p3->extra[j];
                for (int k = 0; k < ORC_VS_INPUTS; k++) smix_fr(&g, S(&in->inputs[k]));
                for (int c = 0; c < 2; c++)
                    for (int d = 0; d < VS_DEG; d++) smix_fr(&g, &xpoly[((i * EX) + 2 * j + c) * VS_DEG + d]);
                poly_eval(&xtau[i * EX + 2 * j + 1], &xpoly[(i * EX + 2 * j + 1) * VS_DEG], &tau);
                fr_t xh2, zn2, t2;
                vs_input_eval(&xh2, &zn2, (const fr_t *)in->inputs, &tau);
                fr_mul(&t2, &zn2, &xtau[i * EX + 2 * j + 1]);
                fr_add(&xtau[i * EX + 2 * j], &xh2, &t2);
                for (int c = 0; c < 2; c++) {
                    if (sh->hiding) smix_fr(&g, &xrho[i * EX + 2 * j + c]);
                    fr_mul(&t2, &gam, &xrho[i * EX + 2 * j + c]);
                    fr_add(&xcval[i * EX + 2 * j + c], &xtau[i * EX + 2 * j + c], &t2);
                }
            }
            if (MC > 1) { /* the proof's circuits: ids ascending, each with at least one instance */
                orc_vs_proof4 *p4 = REC4(i);
                const uint32_t ni = p3->n_instances;
                const uint32_t lim = (uint32_t)MC < ni ? (uint32_t)MC : ni;
                const uint32_t nc = 1 + (uint32_t)(smix_next(&g) % lim);
                p4->n_circuits = nc;
                p->circuit = (uint32_t)(smix_next(&g) % (n_circuits - (nc - 1)));
                uint32_t c_prev = p->circuit, f_prev = 0;
                for (uint32_t d = 0; d + 1 < nc; d++) {
                    const uint32_t rem = nc - 2 - d; /* parts still to come after this one */
                    orc_vs_circuit_part *cp = &p4->more[d];
                    cp->circuit = c_prev + 1 + (uint32_t)(smix_next(&g) % (n_circuits - 1 - rem - c_prev));
                    cp->first_instance = f_prev + 1 + (uint32_t)(smix_next(&g) % (ni - 1 - rem - f_prev));
                    c_prev = cp->circuit;
                    f_prev = cp->first_instance;
                }
            }
        }
    }
    orc_g1_affine *pcomm = (orc_g1_affine *)malloc(sizeof(orc_g1_affine) * n * PC * 2);
    orc_g1_fixed_base_batch(pcomm, (const orc_fr *)cval, n * PC * 2);
    for (size_t i = 0; i < n; i++)
        for (size_t k = 0; k < PC; k++) {
            REC2(i)->base.comms[k] = pcomm[(i * PC + k) * 2];
            const int c = sh->bound_class[k];
            if (c >= 0 && c < sh->n_bounds) REC2(i)->shifted[k] = pcomm[(i * PC + k) * 2 + 1];
        }
    if (MI > 1) {
        orc_g1_affine *xcomm = (orc_g1_affine *)malloc(sizeof(orc_g1_affine) * n * EX);
        orc_g1_fixed_base_batch(xcomm, (const orc_fr *)xcval, n * EX);
        for (size_t i = 0; i < n; i++)
            for (uint32_t j = 0; j + 1 < REC3(i)->n_instances; j++)
                for (int c = 0; c < 2; c++) REC3(i)->extra[j].comms[c] = xcomm[i * EX + 2 * j + c];
        free(xcomm);
    }
    fr_t *wk = (fr_t *)malloc(sizeof(fr_t) * n * 2);
    for (size_t i = 0; i < n; i++) {
        orc_vs_proof2 *p2 = REC2(i);
        orc_vs_proof *p = &p2->base;
        const orc_vs_circuit_vk *cvk = &circuits[p->circuit];
        const int m1 = MI > 1 ? (int)REC3(i)->n_instances - 1 : 0;
        orc_vs_instance *inst = MI > 1 ? REC3(i)->extra : NULL;
        vs_parts pt = {NULL, 0, circuits};
        if (MC > 1) {
            pt.parts = REC4(i)->more;
            pt.d1 = (int)REC4(i)->n_circuits - 1;
        }
        fr_t z[2], xi, xip, t, num[2], rv[2];
        vs_challenges_pt(cvk, p, (const uint8_t *)inst, m1, &pt, (orc_fr *)&z[0], (orc_fr *)&z[1], (orc_fr *)&xi);
        for (int d = 0; d < pt.d1; d++) /* the further circuits' polynomials at their query points */
            for (int k = 0; k < ORC_VS_VK_COMMS; k++)
                poly_eval(S(&REC4(i)->more[d].vk_evals[k]), &cpoly[(REC4(i)->more[d].circuit * ORC_VS_VK_COMMS + k) * VS_DEG],
                          &z[sh->query_of[ORC_VS_PROOF_COMMS + k] & 1]);
        for (int k = 0; k < ORC_VS_EVALS; k++) {
            const int q = sh->query_of[k] & 1;
            const fr_t *poly = k < ORC_VS_PROOF_COMMS ? &ppoly[(i * PC + k) * VS_DEG]
                                                      : &cpoly[(p->circuit * ORC_VS_VK_COMMS + (k - ORC_VS_PROOF_COMMS)) * VS_DEG];
            poly_eval(S(&p->evals[k]), poly, &z[q]);
        }
        { /* p_0(z_beta) = x_hat(z_beta) + (z_beta^16-1) p_1(z_beta): p_0 and p_1 must open at the beta query */
            fr_t xh, znm1;
            if (vs_input_eval(&xh, &znm1, (const fr_t *)p->inputs, &z[0])) {
                fr_mul(&t, &znm1, S(&p->evals[1]));
                fr_add(S(&p->evals[0]), &xh, &t);
            }
        }
        for (int j = 0; j < m1; j++) { /* b_j = B_j(z_beta), a_j from the instance's identity */
            fr_t xh, znm1;
            poly_eval(S(&inst[j].evals[1]), &xpoly[(i * EX + 2 * j + 1) * VS_DEG], &z[0]);
            if (vs_input_eval(&xh, &znm1, (const fr_t *)inst[j].inputs, &z[0])) {
                fr_mul(&t, &znm1, S(&inst[j].evals[1]));
                fr_add(S(&inst[j].evals[0]), &xh, &t);
            }
        }
        vs_challenges_pt(cvk, p, (const uint8_t *)inst, m1, &pt, (orc_fr *)&z[0], (orc_fr *)&z[1], (orc_fr *)&xi);
        fr_one(&xip);
        for (int q = 0; q < 2; q++) { fr_zero(&num[q]); fr_zero(&rv[q]); }
        fr_t kap[8];
        (void)vs2_lc_factors(kap, z, need_inv);
        for (int k = 0; k < ORC_VS_EVALS; k++) {
            const int q = sh->query_of[k] & 1;
            const fr_t *at_tau = k < ORC_VS_PROOF_COMMS ? &ptau[i * PC + k] : &ctau[p->circuit * ORC_VS_VK_COMMS + (k - ORC_VS_PROOF_COMMS)];
            fr_t coef;
            fr_mul(&coef, &xip, &kap[sh->lc_kind[k] & 7]);
            fr_sub(&t, at_tau, S(&p->evals[k]));
            fr_mul(&t, &t, &coef);
            fr_add(&num[q], &num[q], &t);
            if (k < ORC_VS_PROOF_COMMS) {
                fr_mul(&t, &coef, &prho[i * PC + k]);
                fr_add(&rv[q], &rv[q], &t);
            }
            fr_mul(&xip, &xip, &xi);
        }
        if (m1) {
            for (int j = 0; j < ORC_VS_PROOF_COMMS; j++) fr_mul(&xip, &xip, &xi); /* past the degree-bound terms: xi^30 */
            for (int j = 0; j < m1; j++)
                for (int c = 0; c < 2; c++) {
                    fr_sub(&t, &xtau[i * EX + 2 * j + c], S(&inst[j].evals[c]));
                    fr_mul(&t, &t, &xip);
                    fr_add(&num[0], &num[0], &t);
                    fr_mul(&t, &xip, &xrho[i * EX + 2 * j + c]);
                    fr_add(&rv[0], &rv[0], &t);
                    fr_mul(&xip, &xip, &xi);
                }
        }
        if (pt.d1) { /* xip = xi^(30 + 2 m1) -> xi^36, then the further circuits' polynomials (never hiding) */
            for (int j = m1; j < ORC_VS_MAX_INSTANCES - 1; j++) { fr_mul(&xip, &xip, &xi); fr_mul(&xip, &xip, &xi); }
            for (int d = 0; d < pt.d1; d++)
                for (int k = 0; k < ORC_VS_VK_COMMS; k++) {
                    const int pp = ORC_VS_PROOF_COMMS + k, q = sh->query_of[pp] & 1;
                    fr_t coef;
                    fr_mul(&coef, &xip, &kap[sh->lc_kind[pp] & 7]);
                    fr_sub(&t, &ctau[pt.parts[d].circuit * ORC_VS_VK_COMMS + k], S(&pt.parts[d].vk_evals[k]));
                    fr_mul(&t, &t, &coef);
                    fr_add(&num[q], &num[q], &t);
                    fr_mul(&xip, &xip, &xi);
                }
        }
        for (int q = 0; q < 2; q++) {
            fr_sub(&t, &tau, &z[q]);
            fr_inv(&t, &t);
            fr_mul(&wk[2 * i + q], &num[q], &t);
            *S(&p2->random_v[q]) = rv[q];
        }
    }
    orc_g1_affine *wc = (orc_g1_affine *)malloc(sizeof(orc_g1_affine) * n * 2);
    orc_g1_fixed_base_batch(wc, (const orc_fr *)wk, n * 2);
    for (size_t i = 0; i < n; i++) {
        REC2(i)->base.w[0] = wc[2 * i];
        REC2(i)->base.w[1] = wc[2 * i + 1];
        expected[i] = 1;
    }
    for (size_t i = 0; i < n; i++) {
        if ((smix_next(&g) & 1023) >= corrupt_per_1024) continue;
        orc_vs_proof2 *p2 = REC2(i);
        orc_vs_proof *p = &p2->base;
        fr_t one;
        fr_one(&one);
        if (MC > 1 && REC4(i)->n_circuits > 1 && (smix_next(&g) & 3) == 0) { /* spoil a further circuit's part instead */
            orc_vs_circuit_part *cp = &REC4(i)->more[smix_next(&g) % (REC4(i)->n_circuits - 1)];
            switch ((int)(smix_next(&g) % 4)) {
            case 0: { int k = (int)(smix_next(&g) % ORC_VS_VK_COMMS); fr_add(S(&cp->vk_evals[k]), S(&cp->vk_evals[k]), &one); break; }
            case 1: cp->circuit = p->circuit; break;                 /* ids must ascend strictly */
            case 2: cp->first_instance = REC3(i)->n_instances; break; /* a circuit without an instance */
            default: REC4(i)->n_circuits = (uint32_t)MC + 1; break;  /* more circuits than the shape allows */
            }
            expected[i] = 0;
            continue;
        }
        if (MI > 1 && REC3(i)->n_instances > 1 && (smix_next(&g) & 1)) { /* spoil a further instance instead */
            orc_vs_instance *in = &REC3(i)->extra[smix_next(&g) % (REC3(i)->n_instances - 1)];
            switch ((int)(smix_next(&g) % 3)) {
            case 0: fr_add(S(&in->evals[1]), S(&in->evals[1]), &one); break;
            case 1: fr_add(S(&in->inputs[smix_next(&g) % ORC_VS_INPUTS]), S(&in->inputs[1]), &one); break;
            default: in->comms[smix_next(&g) & 1] = vk->g; break;
            }
            expected[i] = 0;
            continue;
        }
        int kind = (int)(smix_next(&g) % 6);
        if (kind == 4 && !sh->hiding) kind = 0;
        if (kind == 5 && !sh->n_bounds) kind = 1;
        switch (kind) {
        case 0: { int k = 2 + (int)(smix_next(&g) % (ORC_VS_EVALS - 2)); fr_add(S(&p->evals[k]), S(&p->evals[k]), &one); break; }
        case 1: p->w[smix_next(&g) & 1] = vk->g; break;
        case 2: fr_add(S(&p->inputs[smix_next(&g) % ORC_VS_INPUTS]), S(&p->inputs[0]), &one); break;
        case 3: p->comms[2 + smix_next(&g) % 7] = vk->g; break;
        case 4: { int q = (int)(smix_next(&g) & 1); fr_add(S(&p2->random_v[q]), S(&p2->random_v[q]), &one); break; }
        default: {
            int j = 0;
            for (int t2 = 0; t2 < ORC_VS_PROOF_COMMS; t2++)
                if (sh->bound_class[t2] >= 0 && sh->bound_class[t2] < sh->n_bounds) j = t2; /* the last bounded polynomial */
            p2->shifted[j] = vk->g;
            break;
        }
        }
        expected[i] = 0;
    }
    free(cpoly); free(ctau); free(ccomm); free(ppoly); free(ptau); free(prho); free(cval); free(pcomm); free(wk); free(wc);
    free(xpoly); free(xtau); free(xrho); free(xcval);
#undef REC2
#undef REC3
#undef REC4
}

/* ------------------------------------------------------------------ wire records (orc.h) */
/* a record as a run of segments: raw bytes | Fr fields | G1 points, each with the class of slot it sits in */
enum { SEG_RAW, SEG_FR, SEG_G1 };
enum { CLS_BASE = 0, CLS_SHIFTED = 1, CLS_INST = 2 /* + j */, CLS_PART = 8 /* + d */ };
typedef struct { int type, count, cls; } vs_seg;
static int vs_layout(int kind, vs_seg *seg) {
    int n = 0;
    seg[n++] = (vs_seg){SEG_RAW, 16, CLS_BASE};
    seg[n++] = (vs_seg){SEG_FR, ORC_VS_INPUTS, CLS_BASE};
    seg[n++] = (vs_seg){SEG_G1, ORC_VS_PROOF_COMMS, CLS_BASE};
    seg[n++] = (vs_seg){SEG_FR, ORC_VS_EVALS, CLS_BASE};
    seg[n++] = (vs_seg){SEG_G1, 2, CLS_BASE};
    if (kind >= 2) {
        seg[n++] = (vs_seg){SEG_FR, 2, CLS_BASE};
        seg[n++] = (vs_seg){SEG_G1, ORC_VS_PROOF_COMMS, CLS_SHIFTED};
    }
    if (kind >= 3) {
        seg[n++] = (vs_seg){SEG_RAW, 16, CLS_BASE};
        for (int j = 0; j < ORC_VS_MAX_INSTANCES - 1; j++) {
            seg[n++] = (vs_seg){SEG_FR, ORC_VS_INPUTS, CLS_INST + j};
            seg[n++] = (vs_seg){SEG_G1, 2, CLS_INST + j};
            seg[n++] = (vs_seg){SEG_FR, 2, CLS_INST + j};
        }
    }
    if (kind >= 4) {
        seg[n++] = (vs_seg){SEG_RAW, 16, CLS_BASE};
        for (int d = 0; d < ORC_VS_MAX_CIRCUITS - 1; d++) {
            seg[n++] = (vs_seg){SEG_RAW, 16, CLS_PART + d};
            seg[n++] = (vs_seg){SEG_FR, ORC_VS_VK_COMMS, CLS_PART + d};
        }
    }
    return n;
}
static size_t vs_wire_bytes_kind(int kind) {
    vs_seg seg[32] = {{0, 0, 0}};
    const int ns = vs_layout(kind, seg);
    size_t b = 0;
    for (int k = 0; k < ns; k++) b += (size_t)seg[k].count * (seg[k].type == SEG_RAW ? 1 : (seg[k].type == SEG_FR ? 32 : 48));
    return b;
}
size_t orc_vs_wire_bytes(const orc_vs_shape *sh) { return vs_wire_bytes_kind(vs3_kind(sh)); }
static int vs_slot_used(const orc_vs_shape *sh, int cls, int j, int m1, int d1) {
    if (cls == CLS_BASE) return 1;
    if (cls == CLS_SHIFTED) return sh && sh->bound_class[j] >= 0 && sh->bound_class[j] < sh->n_bounds;
    if (cls >= CLS_PART) return cls - CLS_PART < d1;
    return cls - CLS_INST < m1;
}
/* the counts a record's headers announce, clamped to what the shape allows (out of range: first instance / circuit only) */
static void vs_counts(const orc_vs_shape *sh, int kind, uint32_t ni, uint32_t nc, int *m1, int *d1) {
    *m1 = *d1 = 0;
    if (kind >= 3) {
        const uint32_t mi = sh->max_instances > 1 ? sh->max_instances : 1;
        if (ni >= 1 && ni <= mi && ni <= ORC_VS_MAX_INSTANCES) *m1 = (int)ni - 1;
    }
    if (kind >= 4 && nc >= 1 && nc <= sh->max_circuits && nc <= ORC_VS_MAX_CIRCUITS) *d1 = (int)nc - 1;
}
void orc_vs_wire_encode(const orc_vs_shape *sh, const void *records, size_t n, uint8_t *wire) {
    const int kind = vs3_kind(sh);
    vs_seg seg[32] = {{0, 0, 0}};
    const int ns = vs_layout(kind, seg);
    const size_t rb = vs3_stride(sh), wb = orc_vs_wire_bytes(sh);
    for (size_t i = 0; i < n; i++) {
        const uint8_t *r = (const uint8_t *)records + rb * i;
        uint8_t *w = wire + wb * i;
        uint32_t ni = 0, nc = 0;
        if (kind >= 3) memcpy(&ni, r + sizeof(orc_vs_proof2), 4);
        if (kind >= 4) memcpy(&nc, r + sizeof(orc_vs_proof3), 4);
        int m1, d1;
        vs_counts(sh, kind, ni, nc, &m1, &d1);
        for (int k = 0; k < ns; k++)
            for (int j = 0; j < (seg[k].type == SEG_RAW ? 1 : seg[k].count); j++) {
                const int used = vs_slot_used(sh, seg[k].cls, j, m1, d1);
                if (seg[k].type == SEG_RAW) {
                    if (used) memcpy(w, r, (size_t)seg[k].count); else memset(w, 0, (size_t)seg[k].count);
                    r += seg[k].count;
                    w += seg[k].count;
                } else if (seg[k].type == SEG_FR) {
                    memset(w, 0, 32);
                    if (used) fr_to_canonical((uint64_t *)w, (const fr_t *)r);
                    r += 32;
                    w += 32;
                } else {
                    memset(w, 0, 48);
                    if (used) orc_g1_compress(w, (const orc_g1_affine *)r);
                    r += 96;
                    w += 48;
                }
            }
    }
}
void orc_vs_wire_decode(const orc_vs_shape *sh, const uint8_t *wire, size_t n, void *records, uint8_t *dec_ok, int threads) {
    const int kind = vs3_kind(sh);
    vs_seg seg[32] = {{0, 0, 0}};
    const int ns = vs_layout(kind, seg);
    const size_t rb = vs3_stride(sh), wb = orc_vs_wire_bytes(sh);
    const size_t off_ni = vs_wire_bytes_kind(2), off_nc = vs_wire_bytes_kind(3); /* where the counts sit in the wire record */
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads)
    for (size_t i = 0; i < n; i++) {
        const uint8_t *w = wire + wb * i;
        uint8_t *r = (uint8_t *)records + rb * i;
        uint32_t ni = 0, nc = 0;
        if (kind >= 3) memcpy(&ni, w + off_ni, 4);
        if (kind >= 4) memcpy(&nc, w + off_nc, 4);
        int m1, d1, ok = 1;
        vs_counts(sh, kind, ni, nc, &m1, &d1);
        for (int k = 0; k < ns; k++)
            for (int j = 0; j < (seg[k].type == SEG_RAW ? 1 : seg[k].count); j++) {
                const int used = vs_slot_used(sh, seg[k].cls, j, m1, d1);
                if (seg[k].type == SEG_RAW) {
                    if (used) memcpy(r, w, (size_t)seg[k].count); else memset(r, 0, (size_t)seg[k].count);
                    r += seg[k].count;
                    w += seg[k].count;
                } else if (seg[k].type == SEG_FR) {
                    memset(r, 0, 32);
                    if (used) {
                        uint64_t c[4];
                        memcpy(c, w, 32);
                        if (limbs_lt(c, ORC_FR_MODULUS, 4)) fr_from_canonical((fr_t *)r, c);
                        else ok = 0;
                    }
                    r += 32;
                    w += 32;
                } else {
                    memset(r, 0, 96);
                    if (used) {
                        orc_g1_affine pt;
                        if (orc_g1_decompress(&pt, w, 1)) memcpy(r, &pt, 96);
                        else ok = 0;
                    }
                    r += 96;
                    w += 48;
                }
            }
        dec_ok[i] = (uint8_t)ok;
    }
}
void orc_vs2_generate(uint64_t seed, size_t n, size_t n_circuits, uint32_t corrupt_per_1024, const orc_vs_shape *sh,
                      orc_kzg_vk *vk, orc_vs_key_ext *ext, orc_vs_circuit_vk *circuits, orc_vs_proof2 *proofs,
                      uint8_t *randomizers /* n x 48 */, uint8_t *expected) {
    orc_vs3_generate(seed, n, n_circuits, corrupt_per_1024, sh, vk, ext, circuits, proofs, randomizers, expected);
}

/* ------------------------------------------------------------------ self test */
int orc_self_test(void) {
    orc_g1_affine g;
    orc_g2_affine h;
    orc_g1_generator(&g);
    orc_g2_generator(&h);
    if (!orc_g1_is_on_curve(&g) || !orc_g2_is_on_curve(&h)) return 1;
    orc_bigint256 rm1;
    memcpy(rm1.l, ORC_FR_MODULUS, 32);
    rm1.l[0] -= 1;
    orc_g1_jac t;
    orc_g1_affine a, ng;
    orc_g1_mul(&t, &g, &rm1);
    orc_g1_to_affine(&a, &t);
    orc_g1_neg_affine(&ng, &g);
    if (memcmp(&a, &ng, sizeof a)) return 2;
    /* bilinearity e(5G, 7H) == e(G, H)^35 */
    orc_bigint256 k5 = {{5, 0, 0, 0}}, k7 = {{7, 0, 0, 0}};
    orc_g2_affine h7;
    orc_g1_affine g5;
    orc_g1_mul(&t, &g, &k5);
    orc_g1_to_affine(&g5, &t);
    orc_g2_mul(&h7, &h, &k7);
    orc_fq12 e1, e0, acc;
    orc_pairing(&e1, &g5, &h7);
    orc_pairing(&e0, &g, &h);
    orc_fq12_one(&acc);
    for (int i = 0; i < 35; i++) orc_fq12_mul(&acc, &acc, &e0);
    if (memcmp(&acc, &e1, sizeof acc)) return 3;
    if (orc_fq12_is_one(&e0)) return 4;
    return 0;
}
