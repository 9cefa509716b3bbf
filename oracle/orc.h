/* CPU oracle for the MI355X batch-SNARK-verification hot path.
 *
 * TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.  The product (snarkos_amd/, libmi355x_snark.so)
 * never links, imports or calls anything in oracle/.
 *
 * PARITY UNPINNED.  The reference (zosorock/snarkOS) reaches this arithmetic only through
 * Ledger::check_next_block / check_transaction_basic
 * (reference node/bft/ledger-service/src/ledger.rs:319-347); the arithmetic itself is in
 * snarkvm-{algorithms,curves,fields} 1.0.0 @ git rev dea322b3d374ac0fa7dae26cf006bf67acd52e9e
 * (reference Cargo.toml:47-52, Cargo.lock:3578-3580), which is not vendored, and the
 * reference's own tests hold no golden vector for this path.  This
 * file restates the published algorithms (Montgomery CIOS, Jacobian G1, ate pairing over the
 * BLS12 seed, Pippenger bucket MSM, radix-2 NTT, KZG10 check / batch_check) and is pinned by
 * oracle/pyref.py (independent Python big-int implementation) and by algebraic identities
 * (tests/test_oracle_*.py), not by reference outputs.
 */
#ifndef ORC_H
#define ORC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- ABI-layout types (identical to include/mi355x_snark.h) ---- */
typedef struct { uint64_t l[4]; } orc_fr;        /* Montgomery unless stated */
typedef struct { uint64_t l[4]; } orc_bigint256; /* canonical integer */
typedef struct { uint64_t l[6]; } orc_fq;        /* Montgomery */
typedef struct { orc_fq x, y; } orc_g1_affine;   /* 96 B; infinity <=> x == y == 0 */
typedef struct { orc_fq x, y, z; } orc_g1_jac;   /* 144 B Jacobian; infinity <=> z == 0 */
typedef struct { orc_fq c0, c1; } orc_fq2;
typedef struct { orc_fq2 x, y; } orc_g2_affine;  /* 192 B; infinity <=> all zero */
typedef struct { orc_fq2 c0, c1, c2; } orc_fq6;
typedef struct { orc_fq6 c0, c1; } orc_fq12;     /* 576 B */
typedef struct { orc_g1_affine g; orc_g2_affine h, tau_h; } orc_kzg_vk;

/* ---- fields ---- */
void orc_fr_mul(orc_fr *r, const orc_fr *a, const orc_fr *b);
void orc_fr_add(orc_fr *r, const orc_fr *a, const orc_fr *b);
void orc_fr_sub(orc_fr *r, const orc_fr *a, const orc_fr *b);
void orc_fr_inv(orc_fr *r, const orc_fr *a);
void orc_fr_from_canonical(orc_fr *r, const orc_bigint256 *a);
void orc_fr_to_canonical(orc_bigint256 *r, const orc_fr *a);
void orc_fq_mul(orc_fq *r, const orc_fq *a, const orc_fq *b);
void orc_fq_add(orc_fq *r, const orc_fq *a, const orc_fq *b);
void orc_fq_sub(orc_fq *r, const orc_fq *a, const orc_fq *b);
void orc_fq_inv(orc_fq *r, const orc_fq *a);
void orc_fq_from_canonical(orc_fq *r, const uint64_t *limbs6);
void orc_fq_to_canonical(uint64_t *limbs6, const orc_fq *a);
void orc_fq12_mul(orc_fq12 *r, const orc_fq12 *a, const orc_fq12 *b);
void orc_fq12_inv(orc_fq12 *r, const orc_fq12 *a);
void orc_fq12_frobenius(orc_fq12 *r, const orc_fq12 *a, int power);
void orc_fq12_one(orc_fq12 *r);
int orc_fq12_is_one(const orc_fq12 *a);

/* ---- G1 / G2 ---- */
void orc_g1_generator(orc_g1_affine *g);
void orc_g2_generator(orc_g2_affine *g);
int orc_g1_is_on_curve(const orc_g1_affine *p);
void orc_g1_add(orc_g1_jac *r, const orc_g1_jac *a, const orc_g1_jac *b);
void orc_g1_add_mixed(orc_g1_jac *r, const orc_g1_jac *a, const orc_g1_affine *b);
void orc_g1_double(orc_g1_jac *r, const orc_g1_jac *a);
void orc_g1_neg_affine(orc_g1_affine *r, const orc_g1_affine *a);
/* compressed points (48 bytes; convention in orc_core.c) */
int orc_fq_sqrt(orc_fq *out, const orc_fq *a);
void orc_g1_compress(uint8_t out[48], const orc_g1_affine *p);
int orc_g1_decompress(orc_g1_affine *out, const uint8_t in[48], int check_subgroup);
void orc_g1_from_affine(orc_g1_jac *r, const orc_g1_affine *a);
void orc_g1_to_affine(orc_g1_affine *r, const orc_g1_jac *a);
void orc_g1_batch_to_affine(orc_g1_affine *r, const orc_g1_jac *a, size_t n);
void orc_g1_mul(orc_g1_jac *r, const orc_g1_affine *p, const orc_bigint256 *k);
void orc_g2_mul(orc_g2_affine *r, const orc_g2_affine *p, const orc_bigint256 *k);
int orc_g2_is_on_curve(const orc_g2_affine *p);
/* k*G for many k with a fixed-base table (generator only; used by the synthetic prover) */
void orc_g1_fixed_base_batch(orc_g1_affine *out, const orc_fr *k_mont, size_t n);

/* ---- MSM (snarkvm_algorithms::msm::VariableBase::msm; scalars canonical) ---- */
void orc_msm_naive(orc_g1_jac *out, const orc_g1_affine *bases, const orc_bigint256 *scalars, size_t n);
void orc_msm_pippenger(orc_g1_jac *out, const orc_g1_affine *bases, const orc_bigint256 *scalars, size_t n,
                       int threads);

/* ---- NTT (snarkvm_algorithms::fft::EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place) ---- */
void orc_fr_omega(orc_fr *w, uint32_t log_n);
int orc_ntt(orc_fr *data, uint32_t log_n, int inverse, int coset, int threads);
void orc_dft_naive(orc_fr *out, const orc_fr *in, uint32_t log_n);

/* ---- pairing ---- */
#define ORC_MILLER_STEPS 70 /* 63 doublings + 6 additions (+1 spare) for x = 0x8508c00000000001 */
typedef struct { orc_fq2 c1, c3; } orc_line;  /* l(P) = yP + (c1 * xP) w + c3 w^3 */
typedef struct { int n; int infinity; orc_line lines[ORC_MILLER_STEPS]; } orc_g2_prepared;
void orc_g2_prepare(orc_g2_prepared *out, const orc_g2_affine *q);
void orc_miller_loop(orc_fq12 *f, const orc_g1_affine *p, const orc_g2_prepared *q);
void orc_final_exponentiation(orc_fq12 *r, const orc_fq12 *f); /* f^(3 (q^12-1)/r) */
void orc_pairing(orc_fq12 *r, const orc_g1_affine *p, const orc_g2_affine *q);
int orc_pairing_product_is_one(const orc_g1_affine *ps, const orc_g2_prepared *const *qs, size_t n);

/* ---- KZG10 (snarkvm_algorithms::polycommit::kzg10::KZG10::{check,batch_check}) ---- */
int orc_kzg_check(const orc_kzg_vk *vk, const orc_g1_affine *comm, const orc_fr *point, const orc_fr *value,
                  const orc_g1_affine *witness);
int orc_kzg_batch_check(const orc_kzg_vk *vk, const orc_g1_affine *comms, const orc_fr *points,
                        const orc_fr *values, const orc_g1_affine *witnesses, size_t n,
                        const uint8_t *randomizers128);
int orc_kzg_batch_check_hiding(const orc_kzg_vk *vk, const orc_g1_affine *gamma_g, const orc_g1_affine *comms,
                               const orc_fr *points, const orc_fr *values, const orc_g1_affine *witnesses,
                               const orc_fr *random_v, size_t n, const uint8_t *randomizers128);

/* prover side: KZG10::commit without hiding (coefficients Montgomery, converted with
 * to_bigint() as the reference does before its msm call); schoolbook polynomial product; Horner evaluation;
 * witness polynomial (p(x) - p(z)) / (x - z) by synthetic division, returns p(z) */
void orc_kzg_commit(orc_g1_jac *out, const orc_g1_affine *powers, const orc_fr *coeffs, size_t n, int threads);
void orc_poly_mul_schoolbook(orc_fr *out, const orc_fr *a, size_t na, const orc_fr *b, size_t nb);
void orc_poly_eval(orc_fr *out, const orc_fr *coeffs, size_t n, const orc_fr *z);
void orc_poly_witness(orc_fr *q, orc_fr *value, const orc_fr *coeffs, size_t n, const orc_fr *z);

/* ---- Poseidon sponge over Fq (crypto_hash::PoseidonSponge<Fq,2,1>) ----
 * width 3, S-box x^17, 4 + 31 + 4 rounds; constants generated by tools/gen_constants.py (snarkVM's are not
 * available offline: parity unpinned).  inputs: n x m Montgomery Fq, outputs: n x k. */
void orc_poseidon_permute(orc_fq state[3]);
void orc_poseidon_hash_many(const orc_fq *inputs, size_t n, size_t m, size_t k, orc_fq *outputs, int threads);

/* ---- SHA-256 (stand-in Fiat-Shamir transcript) ---- */
void orc_sha256(uint8_t out[32], const uint8_t *data, size_t len);

/* ---- VarunaShaped-v1 synthetic batch (DESIGN.md section "VarunaShaped-v1") ---- */
#define ORC_VS_VK_COMMS 12
#define ORC_VS_PROOF_COMMS 9
#define ORC_VS_EVALS 21
#define ORC_VS_INPUTS 16
#define ORC_VS_BETA_PROOF_COMMS 5 /* C_0..C_4 and V_0..V_5 open at beta; the rest at gamma */
#define ORC_VS_BETA_VK_COMMS 6
typedef struct { orc_g1_affine comms[ORC_VS_VK_COMMS]; uint8_t digest[32]; } orc_vs_circuit_vk; /* 1184 B */
typedef struct {
    uint32_t circuit;
    uint32_t _pad[3];
    orc_fr inputs[ORC_VS_INPUTS];
    orc_g1_affine comms[ORC_VS_PROOF_COMMS];
    orc_fr evals[ORC_VS_EVALS];
    orc_g1_affine w[2];
} orc_vs_proof; /* 2256 B */

/* Deterministic synthetic prover with a toy SRS (tau derived from the seed).  corrupt_per_1024 of every
 * 1024 proofs (chosen by the PRNG) are corrupted; expected[i] receives the ground-truth verdict. */
void orc_vs_generate(uint64_t seed, size_t n_proofs, size_t n_circuits, uint32_t corrupt_per_1024,
                     orc_kzg_vk *vk, orc_vs_circuit_vk *circuits, orc_vs_proof *proofs,
                     uint8_t *randomizers128 /* n_proofs*2*16 */, uint8_t *expected);
/* Fiat-Shamir kind for everything below (generate, challenges, verify): 0 SHA-256, 1 Poseidon over Fq */
void orc_vs_set_transcript(int kind);
int orc_vs_get_transcript(void);
int orc_vs_proof_is_well_formed(const orc_vs_proof *p);
/* transcript only: z_beta, z_gamma, xi (Montgomery Fr) */
void orc_vs_challenges(const orc_vs_circuit_vk *cvk, const orc_vs_proof *p, orc_fr *z_beta, orc_fr *z_gamma,
                       orc_fr *xi);
/* one proof at a time, per-proof pairing product (the reference's shape: rayon over transactions) */
int orc_vs_verify_one(const orc_kzg_vk *vk, const orc_g2_prepared *h, const orc_g2_prepared *tau_h,
                      const orc_vs_circuit_vk *circuits, const orc_vs_proof *p, const uint8_t *rand2x16);
void orc_vs_verify_batch_cpu(const orc_kzg_vk *vk, const orc_vs_circuit_vk *circuits, const orc_vs_proof *proofs,
                             size_t n, const uint8_t *randomizers128, uint8_t *verdicts, int threads);
/* aggregated form (one random-linear-combination MSM + one pairing product): returns all_ok */
int orc_vs_verify_aggregated(const orc_kzg_vk *vk, const orc_vs_circuit_vk *circuits, size_t n_circuits,
                             const orc_vs_proof *proofs, size_t n, const uint8_t *randomizers128,
                             uint8_t *alg_ok, orc_g1_affine *total_c, orc_g1_affine *total_w, int threads);

/* ---- VarunaShaped-v2: query-set table, hiding (gamma_g, random_v), degree-bound classes with their own G2 partners
 * (orc_algo.c "VarunaShaped-v2"); three 128-bit randomizers per proof */
#define ORC_VS_MAX_BOUNDS 2
typedef struct {
    uint8_t query_of[ORC_VS_EVALS];          /* 0: the beta query, 1: the gamma query */
    int8_t bound_class[ORC_VS_PROOF_COMMS];  /* degree-bound class of each proof polynomial, -1: none */
    uint8_t hiding, n_bounds;
    /* linear-combination coefficient of each polynomial: its xi-power is further multiplied by
     * (z_beta^16 - 1)^(bit 0) (z_gamma^16 - 1)^(bit 1) -- challenge-dependent coefficients as in Varuna's lincheck
     * combinations (v_H(alpha) v_H(beta) ...); polynomials 0 and 1 take the plain power */
    uint8_t lc_kind[ORC_VS_EVALS];
    /* per-circuit batch size: a proof covers 1 .. max_instances instances of its circuit (0 / 1: one; records are then
     * orc_vs_proof2); with more, records are orc_vs_proof3.  lc_kind bit 2 (VarunaShaped-v3): the coefficient is further
     * multiplied by 1 / (z_beta - z_gamma) */
    uint8_t max_instances;
    /* VarunaShaped-v3: a proof covers instances of 1 .. max_circuits distinct circuits (0 / 1: one); with more, records
     * are orc_vs_proof4 */
    uint8_t max_circuits;
    uint8_t pad[1];
} orc_vs_shape; /* 56 B */
typedef struct { orc_g1_affine gamma_g; orc_g2_affine shift_h[ORC_VS_MAX_BOUNDS]; } orc_vs_key_ext; /* 480 B */
typedef struct { orc_vs_proof base; orc_fr random_v[2]; orc_g1_affine shifted[ORC_VS_PROOF_COMMS]; } orc_vs_proof2; /* 3184 B */
/* a further instance of the proof's circuit: its public inputs, two commitments (the analogues of C_0, C_1), their
 * evaluations at the beta query; a_j = x_hat_j(z_beta) + (z_beta^16 - 1) b_j, and the two polynomials join the beta
 * combination with xi^(30 + 2 (j - 1)), xi^(31 + 2 (j - 1)) */
#define ORC_VS_MAX_INSTANCES 4
typedef struct { orc_fr inputs[ORC_VS_INPUTS]; orc_g1_affine comms[2]; orc_fr evals[2]; } orc_vs_instance; /* 768 B */
typedef struct { orc_vs_proof2 base2; uint32_t n_instances; uint32_t pad[3]; orc_vs_instance extra[ORC_VS_MAX_INSTANCES - 1]; } orc_vs_proof3; /* 5504 B */
/* VarunaShaped-v3: instances of SEVERAL circuits in one proof (Varuna's keys_to_inputs map).  Every circuit after the
 * first brings a part: its id (strictly above the previous circuit's), the index of its first instance (strictly
 * above the previous part's, at least 1, below n_instances) and the evaluations of its 12 polynomials, which open like
 * the first circuit's with xi^(36 + 12 (d - 1) + k) */
#define ORC_VS_MAX_CIRCUITS 3
typedef struct { uint32_t circuit, first_instance, pad[2]; orc_fr vk_evals[ORC_VS_VK_COMMS]; } orc_vs_circuit_part; /* 400 B */
typedef struct { orc_vs_proof3 base3; uint32_t n_circuits; uint32_t pad[3]; orc_vs_circuit_part more[ORC_VS_MAX_CIRCUITS - 1]; } orc_vs_proof4; /* 6320 B */
/* records of the shape's kind (orc_vs_proof2 / 3 / 4); n_circuits = circuits of the context (ids are checked against it) */
void orc_vs4_generate(uint64_t seed, size_t n, size_t n_circuits, uint32_t corrupt_per_1024, const orc_vs_shape *shape,
                      orc_kzg_vk *vk, orc_vs_key_ext *ext, orc_vs_circuit_vk *circuits, void *records,
                      uint8_t *randomizers /* n x 48 */, uint8_t *expected);
/* dec_ok (may be NULL): per-proof flags of a wire decode; a proof whose flag is 0 is invalid */
void orc_vs4_verify_batch_cpu(const orc_kzg_vk *vk, const orc_vs_key_ext *ext, const orc_vs_shape *shape,
                              const orc_vs_circuit_vk *circuits, size_t n_circuits, const void *records, size_t n,
                              const uint8_t *randomizers, const uint8_t *dec_ok, uint8_t *verdicts, int threads);
int orc_vs4_verify_aggregated(const orc_kzg_vk *vk, const orc_vs_key_ext *ext, const orc_vs_shape *shape,
                              const orc_vs_circuit_vk *circuits, size_t n_circuits, const void *records, size_t n,
                              const uint8_t *randomizers, const uint8_t *dec_ok, uint8_t *alg_ok, orc_g1_affine totals[4],
                              int threads);
/* ---- the WIRE form of a record (what the node holds inside Transaction bytes: reference
 * node/router/messages/src/unconfirmed_transaction.rs:25-28, node/bft/events/src/block_response.rs:64-70): every G1
 * point compressed to 48 bytes (orc_g1_compress), every Fr field canonical little-endian, headers as they are; slots a
 * record does not use (shifted commitments of unbounded polynomials, instances / circuits beyond the counts) are zero
 * and are not decoded.  shape == NULL: v1 records (orc_vs_proof).  Sizes: 1728 / 2224 / 4256 / 5072 bytes. */
size_t orc_vs_record_bytes(const orc_vs_shape *shape);
size_t orc_vs_wire_bytes(const orc_vs_shape *shape);
void orc_vs_wire_encode(const orc_vs_shape *shape, const void *records, size_t n, uint8_t *wire);
/* dec_ok[i] = 0 when a used point is not the encoding of a point of the order-r subgroup or a used Fr field is not
 * below r (that field is zero / infinity in the record) */
void orc_vs_wire_decode(const orc_vs_shape *shape, const uint8_t *wire, size_t n, void *records, uint8_t *dec_ok, int threads);
/* the same three entry points over records of the shape's kind (orc_vs_proof2 when max_instances <= 1, else
 * orc_vs_proof3), given as bytes */
void orc_vs3_generate(uint64_t seed, size_t n, size_t n_circuits, uint32_t corrupt_per_1024, const orc_vs_shape *shape,
                      orc_kzg_vk *vk, orc_vs_key_ext *ext, orc_vs_circuit_vk *circuits, void *records,
                      uint8_t *randomizers /* n x 48 */, uint8_t *expected);
void orc_vs3_verify_batch_cpu(const orc_kzg_vk *vk, const orc_vs_key_ext *ext, const orc_vs_shape *shape,
                              const orc_vs_circuit_vk *circuits, const void *records, size_t n,
                              const uint8_t *randomizers, uint8_t *verdicts, int threads);
int orc_vs3_verify_aggregated(const orc_kzg_vk *vk, const orc_vs_key_ext *ext, const orc_vs_shape *shape,
                              const orc_vs_circuit_vk *circuits, const void *records, size_t n,
                              const uint8_t *randomizers, uint8_t *alg_ok, orc_g1_affine totals[4], int threads);
void orc_vs2_generate(uint64_t seed, size_t n, size_t n_circuits, uint32_t corrupt_per_1024, const orc_vs_shape *shape,
                      orc_kzg_vk *vk, orc_vs_key_ext *ext, orc_vs_circuit_vk *circuits, orc_vs_proof2 *proofs,
                      uint8_t *randomizers /* n x 48 */, uint8_t *expected);
void orc_vs2_verify_batch_cpu(const orc_kzg_vk *vk, const orc_vs_key_ext *ext, const orc_vs_shape *shape,
                              const orc_vs_circuit_vk *circuits, const orc_vs_proof2 *proofs, size_t n,
                              const uint8_t *randomizers, uint8_t *verdicts, int threads);
int orc_vs2_verify_aggregated(const orc_kzg_vk *vk, const orc_vs_key_ext *ext, const orc_vs_shape *shape,
                              const orc_vs_circuit_vk *circuits, const orc_vs_proof2 *proofs, size_t n,
                              const uint8_t *randomizers, uint8_t *alg_ok, orc_g1_affine totals[4], int threads);

/* ---- Schnorr signatures over Edwards-BLS12 (orc_schnorr.c; pyref "Schnorr signatures") ----
 * twisted Edwards -x^2 + y^2 = 1 + 3021 x^2 y^2 over Fr; subgroup order s (251 bits), cofactor 4; coordinates Montgomery Fr,
 * scalars canonical integers below s */
typedef struct { orc_fr x, y; } orc_ed_affine; /* 64 B; the identity is (0, 1) */
typedef struct {
    orc_ed_affine address;   /* the signer */
    orc_fr message;          /* one field element: the batch id a certificate's signers sign */
    uint64_t challenge[4], response[4];
    orc_ed_affine pk_sig, pr_sig;
} orc_sig_record; /* 288 B */
void orc_ed_generator(orc_ed_affine *g);
void orc_ed_order(uint64_t s[4]);
int orc_ed_on_curve(const orc_ed_affine *p);
void orc_ed_add(orc_ed_affine *r, const orc_ed_affine *p, const orc_ed_affine *q);
void orc_ed_mul(orc_ed_affine *r, const orc_ed_affine *p, const uint64_t k[4]);
void orc_ed_poseidon_permute(orc_fr *state, int width /* 5 or 9 */);
void orc_ed_hash_to_scalar(uint64_t out[4], const orc_fr *inputs, int n, int rate /* 4 or 8 */);
int orc_schnorr_verify_one(const orc_sig_record *r);
void orc_schnorr_verify_batch(const orc_sig_record *recs, size_t n, uint8_t *verdicts, int threads);
void orc_schnorr_generate(uint64_t seed, size_t n, size_t n_signers, uint32_t corrupt_per_1024, orc_sig_record *recs, uint8_t *expected);

int orc_self_test(void);

#ifdef __cplusplus
}
#endif
#endif
