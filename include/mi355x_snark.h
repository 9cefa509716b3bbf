/* libmi355x_snark -- MI355X-native BLS12-377 batch SNARK verification primitives (C ABI).
 *
 * The drop-in boundary for snarkOS's consensus-side verification path.  snarkOS itself has no FFI
 * seam on this path: it calls Ledger::check_next_block / check_transaction_basic
 * (reference node/bft/ledger-service/src/ledger.rs:319-347, node/consensus/src/lib.rs:499), which run
 * snarkvm_algorithms::{msm, fft, polycommit, snark::varuna} (snarkVM 1.0.0 @ dea322b, the dependency
 * pinned at reference Cargo.toml:47-52).  The entry points below are what a `hip` feature of
 * snarkvm-algorithms would bind, at the same three gates its `cuda` feature uses:
 *
 *   mi_msm_g1*            <- msm::VariableBase::msm(bases: &[G1Affine], scalars: &[BigInteger256])
 *   mi_ntt_fr*            <- fft::EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place
 *   mi_kzg_batch_check    <- polycommit::kzg10::KZG10::batch_check
 *   mi_vs_verify_batch*   <- snark::varuna::VarunaSNARK::verify_batch, driven per block by
 *                            VM::check_transactions; see DESIGN.md "VarunaShaped-v1"
 *                            for the synthetic proof shape this round implements.
 *
 * Conventions
 *   - plain C, no C++/torch types; integers are little-endian 64-bit limbs.
 *   - field elements cross the ABI in Montgomery form (R = 2^256 for Fr, 2^384 for Fq), exactly the
 *     bytes the Rust side holds; MSM scalars are canonical integers < r (BigInteger256).
 *   - mi_g1_affine is 96 bytes {x, y}; the point at infinity is x = y = 0.
 *   - every function returns a status: MI_OK or a negative error class.  A status never carries a
 *     verdict; verdicts are separate uint8 outputs.  On any non-zero status the caller falls back to
 *     its own CPU path (the contract of snarkVM's cuda gate).  No exception or abort crosses the ABI.
 *   - thread-safe and re-entrant: any host thread may call any function; calls are serialised per
 *     device inside the library.  The caller owns every buffer for the duration of the call.
 *   - "_device" variants take HIP device pointers (inputs already resident in HBM) and a hipStream_t
 *     passed as void*; the plain variants take host pointers and stage them.
 *   - one process drives any number of GPUs (snarkOS is one process: reference cli/src/commands/start.rs:616-642):
 *     mi_init names the devices; host-pointer entry points run on the first of them (the primary), "_device" entry
 *     points run on the device that owns their pointers, mi_vs_verify_block shards one block over all of them.
 *   - PRECONDITIONS ON POINTS.  snarkVM validates group elements when it deserialises them; this ABI takes raw
 *     coordinates and so bypasses that step.  The verifier entry points (mi_vs_*) check every G1 point of a proof
 *     for y^2 = x^3 + 1 (or the (0,0) encoding of infinity) and every Fr field for < r, and reject the proof
 *     otherwise.  What NO entry point checks is membership in the prime-order subgroup (cofactor
 *     0x170b5d44300000000000000000000000): callers must pass points of G1 / G2 proper, as snarkVM's deserialiser
 *     guarantees.  mi_msm_g1*, mi_kzg_*, mi_pairing* and mi_bases_register check nothing: their inputs (SRS powers,
 *     verifying keys, the caller's own polynomials) are trusted data; an off-curve point there gives a meaningless
 *     result, never a fault.
 */
#ifndef MI355X_SNARK_H
#define MI355X_SNARK_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_OK 0
#define MI_ERR_BAD_ARGS (-1)
#define MI_ERR_NO_DEVICE (-2)
#define MI_ERR_OOM (-3)
#define MI_ERR_KERNEL (-4)
#define MI_ERR_NOT_INIT (-5)
#define MI_ERR_AGAIN (-6) /* mi_vs_stream_next: the next verdict is not ready; mi_vs_finalize_async: every product stream is busy */
#define MI_ERR_DONE (-7)  /* mi_vs_stream_next: the stream is finished and drained */

typedef struct { uint64_t l[4]; } mi_fr;        /* Fr, Montgomery */
typedef struct { uint64_t l[4]; } mi_bigint256; /* canonical integer < r */
typedef struct { uint64_t l[6]; } mi_fq;        /* Fq, Montgomery */
typedef struct { mi_fq x, y; } mi_g1_affine;    /* 96 B, infinity <=> x == y == 0 */
typedef struct { mi_fq x, y, z; } mi_g1_jac;    /* 144 B Jacobian, infinity <=> z == 0; outputs are normalised (z = 1) */
typedef struct { mi_fq x[2], y[2]; } mi_g2_affine; /* 192 B, Fq2 = c0 + c1 u, u^2 = -5 */
typedef struct { mi_fq c[12]; } mi_fq12;        /* tower order c0.c0.c0, c0.c0.c1, c0.c1.c0, ... (Fq12 = Fq6[w], Fq6 = Fq2[v]) */
typedef struct { mi_g1_affine g; mi_g2_affine h, tau_h; } mi_kzg_vk; /* kzg10::VerifierKey {g, h, beta_h} */

/* ---- lifecycle ---- */
/* SURVEY.md App. C.  device_ids[0] becomes the primary; n_devices == 0 (device_ids may be NULL): the current HIP
 * device.  Idempotent: a second call adds the devices it names that are not in the list yet.  Optional: the first
 * call of any entry point initialises the current device, and a "_device" entry point adds the device of its pointers. */
int32_t mi_init(const int32_t *device_ids, int32_t n_devices);
int32_t mi_device_count(void);          /* devices in the library's list */
int32_t mi_device_id(int32_t index);    /* HIP ordinal of list entry `index` (negative: no such entry) */
int32_t mi_shutdown(void);
const char *mi_error_string(int32_t status);
int32_t mi_abi_version(void);
/* What the embedding process should have arranged and has not (a bit set, never an error; the library sets no
 * environment variable itself: setenv races with a multithreaded caller's getenv).
 *   MI_HINT_HW_QUEUES: GPU_MAX_HW_QUEUES is unset or below 24.  The streamed verifier keeps ~20 HIP streams busy; HIP
 *   maps them onto 4 hardware queues unless the variable says otherwise (a 20-batch stream then takes 11 ms instead of
 *   6).  Set GPU_MAX_HW_QUEUES=32 in the process environment before the first HIP call. */
#define MI_HINT_HW_QUEUES 1
int32_t mi_runtime_hints(void);
/* ---- counters (the build-side analogues of reference node/metrics/src/names.rs:15-48 and of the
 * block / commit latencies recorded at node/consensus/src/lib.rs:511-528) ----  Monotonic since mi_init, summed over
 * all devices; read them twice and take differences for rates (proofs_per_sec = d proofs_verified / d t, ...). */
typedef struct {
    uint64_t batches_verified;     /* batches whose verdict was produced (any entry point) */
    uint64_t proofs_verified;      /* proofs in those batches */
    uint64_t proofs_rejected;      /* proofs named invalid */
    uint64_t verify_batch_nanos;   /* host wall time inside synchronous verify calls + stream submit-to-verdict latencies */
    uint64_t msm_calls, msm_points;            /* mi_msm_g1* / mi_kzg_commit* and the verifier's own MSMs */
    uint64_t ntt_calls, ntt_elements;
    uint64_t pairing_products, pairing_pairs;
    uint64_t points_decompressed;  /* mi_g1_decompress* and wire records */
    uint64_t fallback_batches;     /* batches taken apart proof by proof */
    uint64_t exchange_calls, exchange_nanos; /* all-gathers of partials (any transport) and the host wall time inside them */
    uint64_t reserved[2];
} mi_stats; /* 128 B */
int32_t mi_stats_read(mi_stats *out);

/* ---- msm::VariableBase::msm ---- */
int32_t mi_msm_g1(const mi_g1_affine *bases, const mi_bigint256 *scalars, size_t n, mi_g1_jac *out);
int32_t mi_msm_g1_device(const void *d_bases, const void *d_scalars, size_t n, mi_g1_jac *out, void *stream);
/* sum of n partial results (each a normalised output of mi_msm_g1*: z = 1, or z = 0 for infinity): the combine step
 * of an MSM whose points are split across GPUs; host arithmetic, n is the number of ranks */
int32_t mi_g1_sum(const mi_g1_jac *partials, size_t n, mi_g1_jac *out);
/* bases kept resident in HBM (SRS powers, verifying-key commitments), replicated to every device of the list */
int32_t mi_bases_register(const mi_g1_affine *pts, size_t n, uint64_t *handle_out);
int32_t mi_bases_release(uint64_t handle);
int32_t mi_msm_g1_h(uint64_t bases_handle, size_t offset, const mi_bigint256 *scalars, size_t n, mi_g1_jac *out);

/* ---- fft::EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place ----
 * natural order in and out; omega_n = ROOT^(2^(47 - log_n)); inverse scales by n^-1; coset generator 22 */
int32_t mi_ntt_fr(mi_fr *data, uint32_t log_n, int32_t inverse, int32_t coset);
int32_t mi_ntt_fr_device(void *d_data, uint32_t log_n, int32_t inverse, int32_t coset, void *stream);

/* ---- prover-side reuse of the same kernels ----
 * polycommit::kzg10::KZG10::commit without hiding: sum_i coeffs[i] * powers[offset + i], coefficients in Montgomery
 * form as DensePolynomial holds them (converted to canonical integers on the device, like `to_bigint()` before the
 * reference's msm call); `powers_handle` = mi_bases_register(powers_of_beta_g).  n may be 0 (commitment to zero). */
int32_t mi_kzg_commit_h(uint64_t powers_handle, size_t offset, const mi_fr *coeffs, size_t n, mi_g1_jac *out);
int32_t mi_kzg_commit_h_device(uint64_t powers_handle, size_t offset, const void *d_coeffs, size_t n, mi_g1_jac *out,
                               void *stream);
/* polycommit::kzg10::KZG10::open without hiding: value = p(point) and the commitment to the witness polynomial
 * (p(X) - p(point)) / (X - point), whose coefficients are the suffix Horner sums of p (a scan on the device), committed
 * with the same powers.  coeffs / point / value_out in Montgomery form.  n = 0: value 0, witness infinity. */
int32_t mi_kzg_open_h(uint64_t powers_handle, const mi_fr *coeffs, size_t n, const mi_fr *point, mi_fr *value_out,
                      mi_g1_jac *witness_out);
/* with a hiding polynomial (KZG10::commit / open with hiding_bound = Some(..)): the commitment adds
 * sum rand_coeffs[j] * gamma_powers[j] (powers_of_beta_times_gamma_g, registered as a second handle); the opening proof
 * carries random_v = rho(point) and the witness commits to both quotients */
int32_t mi_kzg_commit_hiding_h(uint64_t powers_handle, uint64_t gamma_powers_handle, const mi_fr *coeffs, size_t n,
                               const mi_fr *rand_coeffs, size_t m, mi_g1_jac *out);
int32_t mi_kzg_open_hiding_h(uint64_t powers_handle, uint64_t gamma_powers_handle, const mi_fr *coeffs, size_t n,
                             const mi_fr *rand_coeffs, size_t m, const mi_fr *point, mi_fr *value_out, mi_fr *random_v_out,
                             mi_g1_jac *witness_out);
/* KZG10::commit_lagrange: evaluations over the domain against the Lagrange-basis SRS (lagrange_handle =
 * mi_bases_register of L_i(beta) G): the same MSM as mi_kzg_commit_h */
int32_t mi_kzg_commit_lagrange_h(uint64_t lagrange_handle, const mi_fr *evals, size_t n, mi_g1_jac *out);
/* EvaluationDomain::divide_by_vanishing_poly_on_coset_in_place: evals[i] = p(g w_M^i), M = 2^log_m (the output of
 * mi_ntt_fr(.., coset = 1)), divided pointwise by Z_N(X) = X^N - 1, N = 2^log_n <= M (log_m - log_n <= 12); a coset
 * inverse NTT of the result gives the quotient's coefficients */
int32_t mi_ntt_divide_by_vanishing_on_coset(mi_fr *evals, uint32_t log_m, uint32_t log_n);
int32_t mi_ntt_divide_by_vanishing_on_coset_device(void *d_evals, uint32_t log_m, uint32_t log_n, void *stream);
/* fft::DensePolynomial multiplication through the evaluation domain of size >= na + nb - 1 (two forward NTTs, a
 * pointwise product, one inverse NTT): out[0 .. na + nb - 1).  na, nb >= 1. */
int32_t mi_poly_mul_fr(const mi_fr *a, size_t na, const mi_fr *b, size_t nb, mi_fr *out);

/* ---- pairing (Bls12_377::{miller_loop, final_exponentiation, product_of_pairings}) ---- */
int32_t mi_pairing_product_is_one(const mi_g1_affine *ps, const mi_g2_affine *qs, size_t n, uint8_t *is_one);
int32_t mi_pairing(const mi_g1_affine *p, const mi_g2_affine *q, mi_fq12 *out); /* f^(3(q^12-1)/r) */

/* ---- kzg10::KZG10::batch_check ----  randomizers128: n x 16 bytes from the caller's RNG */
int32_t mi_kzg_batch_check(const mi_kzg_vk *vk, const mi_g1_affine *comms, const mi_fr *points, const mi_fr *values,
                           const mi_g1_affine *witnesses, size_t n, const uint8_t *randomizers128, uint8_t *ok_out);
/* the same with hiding (VerifierKey.gamma_g, KZGProof.random_v): total_c also carries -(sum r_i random_v[i]) gamma_g.
 * gamma_g and random_v both NULL: identical to mi_kzg_batch_check */
int32_t mi_kzg_batch_check_hiding(const mi_kzg_vk *vk, const mi_g1_affine *gamma_g, const mi_g1_affine *comms,
                                  const mi_fr *points, const mi_fr *values, const mi_g1_affine *witnesses, const mi_fr *random_v,
                                  size_t n, const uint8_t *randomizers128, uint8_t *ok_out);

/* ---- VarunaShaped-v1 batch verification ---- */
#define MI_VS_VK_COMMS 12
#define MI_VS_PROOF_COMMS 9
#define MI_VS_EVALS 21
#define MI_VS_INPUTS 16
typedef struct { mi_g1_affine comms[MI_VS_VK_COMMS]; uint8_t digest[32]; } mi_vs_circuit_vk; /* 1184 B */
typedef struct {
    uint32_t circuit;
    uint32_t _pad[3];
    mi_fr inputs[MI_VS_INPUTS];
    mi_g1_affine comms[MI_VS_PROOF_COMMS];
    mi_fr evals[MI_VS_EVALS];
    mi_g1_affine w[2];
} mi_vs_proof; /* 2256 B */
/* ---- VarunaShaped-v2: the shape of SonicKZG10::check_combinations on top of v1 ----
 *   - the query set is a table (which of the two query points each polynomial opens at), and so are the linear
 *     combinations' coefficients (lc_kind: plain challenge powers, or powers times vanishing-polynomial values at the
 *     challenges);
 *   - hiding: commitments carry rho gamma_g, the proof carries random_v per query (kzg10::VerifierKey.gamma_g,
 *     KZGProof.random_v);
 *   - degree bounds: proof polynomial j of class c comes with a shifted commitment, and e(C_j, shift_h[c]) = e(S_j, h) is
 *     folded into the product with rho_j = r_3 xi^(21 + j): 2 + n_bounds pairs
 *     (total_c - sum rho_j S_j, h), (-total_w, tau h), (sum_{j in c} rho_j C_j, shift_h[c]);
 *   - per-circuit batch sizes: with max_instances > 1 a proof carries 1 .. max_instances instances of its circuit;
 *   - records are mi_vs_proof2, 3184 B (mi_vs_proof3, 5504 B, with instances), and THREE 128-bit randomizers per
 *     proof (48 B).
 * A context made by mi_vs_ctx_create_v2 takes such records through every mi_vs_* entry point (proof pointers are then
 * pointers to mi_vs_proof2; the host-pointer prototypes keep their v1 pointer type, cast). */
#define MI_VS_MAX_BOUNDS 2
typedef struct {
    uint8_t query_of[MI_VS_EVALS];         /* 0: the beta query, 1: the gamma query; polynomials 0 and 1 open at beta */
    int8_t bound_class[MI_VS_PROOF_COMMS]; /* degree-bound class of each proof polynomial, -1: none */
    uint8_t hiding, n_bounds;
    /* linear-combination table: polynomial p's coefficient xi^p is further multiplied by (z_beta^16 - 1)^(bit 0) and
     * (z_gamma^16 - 1)^(bit 1) -- challenge-dependent coefficients, as Varuna's lincheck combinations carry
     * v_H(alpha) v_H(beta)-style factors; polynomials 0 and 1 (the public-input identity) take the plain power */
    uint8_t lc_kind[MI_VS_EVALS];
    /* per-circuit batch size (Varuna's batch_sizes): a proof covers 1 .. max_instances instances of its circuit.
     * 0 / 1: one, records are mi_vs_proof2; more: records are mi_vs_proof3.
     * VarunaShaped-v3: lc_kind bit 2 -- the coefficient is further multiplied by 1 / (z_beta - z_gamma) (coefficients that
     * need the inverse of a challenge; z_beta == z_gamma is an invalid proof under a shape that uses the bit) */
    uint8_t max_instances;
    /* VarunaShaped-v3: a proof covers instances of 1 .. max_circuits DISTINCT circuits (Varuna's keys_to_inputs map);
     * 0 / 1: one; more: records are mi_vs_proof4.  max_circuits <= max_instances. */
    uint8_t max_circuits;
    uint8_t pad[1];
} mi_vs_shape; /* 56 B */
typedef struct { mi_g1_affine gamma_g; mi_g2_affine shift_h[MI_VS_MAX_BOUNDS]; } mi_vs_key_ext; /* 480 B */
typedef struct { mi_vs_proof base; mi_fr random_v[2]; mi_g1_affine shifted[MI_VS_PROOF_COMMS]; } mi_vs_proof2; /* 3184 B */
/* a further instance of the proof's circuit: its public inputs, two commitments A_j, B_j (the analogues of C_0, C_1)
 * and their evaluations a_j, b_j at the beta query: a_j = x_hat_j(z_beta) + (z_beta^16 - 1) b_j, and the two
 * polynomials join the beta combination with xi^(30 + 2 (j - 1)), xi^(31 + 2 (j - 1)).  The transcript chains every
 * further instance in after C_1 (inputs, commitments) and after the evaluations (its two). */
#define MI_VS_MAX_INSTANCES 4
#define MI_VS_EXTRA_INSTANCES 3 /* MI_VS_MAX_INSTANCES - 1 */
typedef struct { mi_fr inputs[MI_VS_INPUTS]; mi_g1_affine comms[2]; mi_fr evals[2]; } mi_vs_instance; /* 768 B */
typedef struct {
    mi_vs_proof2 base2;
    uint32_t n_instances; /* 1 .. the shape's max_instances; anything else: the proof is invalid */
    uint32_t pad[3];
    mi_vs_instance extra[MI_VS_EXTRA_INSTANCES];
} mi_vs_proof3; /* 5504 B */
/* VarunaShaped-v3 (VarunaSNARK::verify_batch takes keys_to_inputs: BTreeMap<&CircuitVerifyingKey,
 * &[Vec<Fr>]>): one proof over instances of SEVERAL circuits.  Instance 0 belongs to the proof's first circuit
 * (base.circuit); every further circuit brings a part: its id (strictly above the previous circuit's: a BTreeMap's key
 * order), the index of its first instance (strictly above the previous part's, >= 1, < n_instances: instances are
 * grouped by circuit) and the evaluations of its 12 polynomials, which open like the first circuit's (the shape's query
 * and linear-combination tables at 9 + k) with xi^(36 + 12 (d - 1) + k); its commitments' scalars merge into that
 * circuit's slots of the aggregate.  Both transcripts absorb every part's header and circuit digest after the first
 * circuit's digest and its evaluations after the instances' (Poseidon: 14 elements, 7 permutations per part).  A count
 * or an id out of range is an invalid proof. */
#define MI_VS_MAX_CIRCUITS 3
#define MI_VS_EXTRA_CIRCUITS 2 /* MI_VS_MAX_CIRCUITS - 1 */
typedef struct { uint32_t circuit, first_instance, pad[2]; mi_fr vk_evals[MI_VS_VK_COMMS]; } mi_vs_circuit_part; /* 400 B */
typedef struct {
    mi_vs_proof3 base3;
    uint32_t n_circuits; /* 1 .. min(the shape's max_circuits, n_instances) */
    uint32_t pad[3];
    mi_vs_circuit_part more[MI_VS_EXTRA_CIRCUITS];
} mi_vs_proof4; /* 6320 B */
/* ---- WIRE records: what the node holds are Transaction bytes (reference
 * node/bft/ledger-service/src/ledger.rs:326-331, node/router/messages/src/unconfirmed_transaction.rs:25-28,
 * node/bft/events/src/block_response.rs:64-70) in which every commitment is a 48-byte compressed point and every field
 * element its canonical little-endian value.  A context created with MI_VS_WIRE takes its records in that form: the
 * record of its kind (mi_vs_proof / 2 / 3 / 4) with every mi_g1_affine replaced by its compressed encoding
 * (mi_g1_compress), every mi_fr by the canonical integer, the 16-byte headers as they are: 1728 / 2224 / 4256 / 5072
 * bytes.  Slots the record does not use (shifted commitments of polynomials without a degree bound, instances and
 * circuits beyond the record's counts) are zero and are not decoded.  The lane pass decodes on the device ahead of the
 * transcript (square root in Fq, sign, and the exact subgroup test snarkVM's deserialiser performs; Fr below r); a point
 * that is not the encoding of a point of the order-r subgroup or a non-canonical Fr makes the proof invalid -- the same
 * verdict as the oracle's.  mi_vs_strides tells a caller the record and randomizer sizes of a context;
 * mi_vs_wire_encode (host arithmetic) turns uncompressed records into wire records. */
#define MI_VS_WIRE 2
/* one shard's accumulator (592 B); bound[c] is infinity for contexts without that class */
typedef struct { mi_g1_jac total_c, total_w, bound[MI_VS_MAX_BOUNDS]; uint64_t n_proofs, n_alg_fail; } mi_vs_partial;

/* keys resident on the device: KZG vk (G2 lines prepared once) + circuit verifying keys */
int32_t mi_vs_ctx_create(const mi_kzg_vk *vk, const mi_vs_circuit_vk *circuits, size_t n_circuits, uint64_t *ctx_out);
/* flags: MI_VS_FS_POSEIDON -- the Fiat-Shamir transcript is a Poseidon sponge over Fq (row A9: width 3, x^17, 8 + 31
 * rounds, this library's own constants) with the reference's absorb volume, 37 permutations per proof (tag, circuit
 * digest, the 16 public inputs as two 128-bit limbs each, the coordinates of C_0..C_8, the 21 evaluations; z_beta,
 * z_gamma, xi = low 128 bits of the first rate element after the 22nd, 26th and 37th permutation), instead of the
 * SHA-256 stand-in of mi_vs_ctx_create.  Proofs are bound to their transcript kind. */
#define MI_VS_FS_POSEIDON 1
/* flags may also carry MI_VS_WIRE (above): the context then takes wire records */
int32_t mi_vs_ctx_create_ex(const mi_kzg_vk *vk, const mi_vs_circuit_vk *circuits, size_t n_circuits, uint32_t flags,
                            uint64_t *ctx_out);
/* VarunaShaped-v2 context: ext may be NULL when the shape neither hides nor bounds; flags as above */
int32_t mi_vs_ctx_create_v2(const mi_kzg_vk *vk, const mi_vs_key_ext *ext, const mi_vs_shape *shape,
                            const mi_vs_circuit_vk *circuits, size_t n_circuits, uint32_t flags, uint64_t *ctx_out);
int32_t mi_vs_ctx_destroy(uint64_t ctx);
/* bytes per proof record and per proof's randomizers that every mi_vs_* entry point of this context reads (host-pointer
 * forms copy n * proof_bytes and n * rand_bytes) */
int32_t mi_vs_strides(uint64_t ctx, size_t *proof_bytes, size_t *rand_bytes);
/* uncompressed records of the context's kind -> wire records (n * the MI_VS_WIRE stride of that kind); works for
 * contexts with and without MI_VS_WIRE */
int32_t mi_vs_wire_encode(uint64_t ctx, const void *records, size_t n, uint8_t *wire_out);
/* host-pointer form: verdicts[n] (1 = valid) and *all_ok.  randomizers128: n x 2 x 16 bytes. */
int32_t mi_vs_verify_batch(uint64_t ctx, const mi_vs_proof *proofs, size_t n, const uint8_t *randomizers128,
                           uint8_t *verdicts, uint8_t *all_ok);
/* device-pointer form (proofs and randomizers already in HBM) */
int32_t mi_vs_verify_batch_device(uint64_t ctx, const void *d_proofs, size_t n, const void *d_randomizers128,
                                  uint8_t *verdicts, uint8_t *all_ok, void *stream);
/* the verdict vector of a batch already known to fail (all_ok = 0 from mi_vs_finalize* or mi_vs_stream_next, which keep
 * no per-proof state): re-derives the scalars and names the invalid proofs without the aggregate MSMs and product.
 * The fallback, here and inside mi_vs_verify_batch*: every term scalar * point (GLV half-length chains), the proofs'
 * accumulators, one pairing product per group of 4 proofs, then per proof inside failing groups; all on the device. */
int32_t mi_vs_identify_device(uint64_t ctx, const void *d_proofs, size_t n, const void *d_randomizers128, uint8_t *verdicts,
                              void *stream);
/* one block-validation batch over ALL devices of the process (SURVEY.md 8e inside one process): contiguous shards, one
 * host thread per device (stage, derive, both MSMs, fold), the 592-byte partials all-gathered over RCCL (grouped
 * ncclAllGather over an in-process communicator; added on the host when RCCL is unavailable or MI_SNARK_EXCHANGE=host),
 * ONE pairing product on the primary; on failure every device names the invalid proofs of its shard.
 * verdicts[n] in block order.  The _device form takes one (proofs, randomizers, count) triple per shard, each already
 * resident on the device that is to verify it. */
int32_t mi_vs_verify_block(uint64_t ctx, const mi_vs_proof *proofs, size_t n, const uint8_t *randomizers128,
                           uint8_t *verdicts, uint8_t *all_ok);
int32_t mi_vs_verify_block_device(uint64_t ctx, const void *const *d_proofs, const void *const *d_randomizers128,
                                  const size_t *n_per_shard, int32_t n_shards, uint8_t *verdicts, uint8_t *all_ok);
/* multi-GPU, one process per GPU: each rank accumulates its shard, partials are exchanged (RCCL all-gather), any rank finalises */
int32_t mi_vs_accumulate_device(uint64_t ctx, const void *d_proofs, size_t n, const void *d_randomizers128,
                                mi_vs_partial *partial_out, void *stream);
int32_t mi_vs_finalize(uint64_t ctx, const mi_vs_partial *partials, size_t n_partials, uint8_t *all_ok);
#define MI_VS_MAX_OPEN_ACCUMULATIONS 14 /* mi_vs_accumulate_begin without a matching _end: 12 by default, env
                                           MI_SNARK_VS_LANES = 1..14, read when a device is first used.  Lanes and
                                           product slots (streams, workspaces) belong to the DEVICE and are shared by
                                           all verifier contexts: the limits hold across contexts.  A begin (and
                                           every call built on it) WAITS for a lane when all are taken. */
#define MI_VS_MAX_TICKETS 16            /* mi_vs_finalize_async without a matching _wait: 8 by default, env
                                           MI_SNARK_VS_SLOTS = 1..16.  Lanes + slots are HIP streams: keep their sum
                                           at 20 or below (the pipeline loses half its throughput when streams have
                                           to share hardware queues) */
/* pipelined form: the pairing product of batch i runs on the context's own stream while the caller accumulates
 * batch i+1 (VM::check_transactions walks a block in chunks of <= 1000 executions; mempool admission streams).
 * At most 8 (MI_SNARK_VS_SLOTS) tickets may be outstanding (each product has its own stream); all must be waited for before the
 * context is destroyed. */
/* split accumulate: _begin enqueues the shard's kernels and returns at once, _end waits and folds on the host.
 * Up to 12 (MI_SNARK_VS_LANES) may be open at a time (each has its own stream and workspaces), so the next batches run on the GPU
 * while batch k is folded and finalised.  Thread safety: every entry point may be called from any host thread; the
 * context lock is dropped around host-side waits and folds, so one thread can enqueue (_begin) while others fold
 * (_end, any order) and a third finalises (set GPU_MAX_HW_QUEUES=32: HIP then maps streams onto 24 hardware queues; keep lanes + products + the
 * process's other streams within that, or a 1.6 ms pairing kernel ends up queued in front of the next MSMs). */
int32_t mi_vs_accumulate_begin(uint64_t ctx, const void *d_proofs, size_t n, const void *d_randomizers128, void *stream,
                               uint64_t *handle_out);
int32_t mi_vs_accumulate_end(uint64_t ctx, uint64_t handle, mi_vs_partial *partial_out);
/* Several batches in ONE lane pass: all batches' derive, merge (and Poseidon transcript) kernels are single launches
 * and all their MSMs are problems of one launch sequence; each batch keeps its own randomizers, accumulators and
 * partial.  A lane pass is a chain of latency-bound kernels and the lanes are few (hardware queues): under load more
 * batches per pass is what raises the throughput.  A launch sequence carries 8 MSMs -- a batch brings two, plus one per
 * degree-bound class -- and all batches must fit the small-MSM path together (2^17 points); otherwise
 * MI_ERR_BAD_ARGS: run them as separate passes.  mi_vs_batches_per_pass tells how many batches of n proofs fit (1..4: eight MSM problems).
 * All batches' inputs must be ready on `stream`.  _end_multi writes partials_out[k] for batch k; `count` must be the
 * pass's own.  _pair = _multi with two batches. */
#define MI_VS_MAX_BATCHES_PER_PASS 8
int32_t mi_vs_accumulate_begin_multi(uint64_t ctx, int32_t count, const void *const *d_proofs, const size_t *n,
                                     const void *const *d_randomizers128, void *stream, uint64_t *handle_out);
int32_t mi_vs_accumulate_end_multi(uint64_t ctx, uint64_t handle, mi_vs_partial *partials_out, int32_t count);
/* The batches of a pass folded into ONE aggregate: the randomizers of all their proofs are independent, so the union of
 * 2..8 batches is one batch-verification instance -- one MSM pair, one host fold, one pairing product for all of them.
 * Ends with mi_vs_accumulate_end: ONE partial over all the proofs (n_proofs = their sum).  A product of one says every
 * batch is valid; otherwise the caller checks the batches on their own (mi_vs_verify_batch_device / mi_vs_identify_device
 * on the still-resident inputs).  MI_ERR_BAD_ARGS when the sum does not fit one pass: use _begin_multi or single passes.
 * The native stream driver does this under load when it has no exchange between ranks (ranks would have to merge the
 * same batches). */
int32_t mi_vs_accumulate_begin_merged(uint64_t ctx, int32_t count, const void *const *d_proofs, const size_t *n,
                                      const void *const *d_randomizers128, void *stream, uint64_t *handle_out);
int32_t mi_vs_accumulate_begin_pair(uint64_t ctx, const void *d_proofs_a, size_t n_a, const void *d_randomizers128_a,
                                    const void *d_proofs_b, size_t n_b, const void *d_randomizers128_b, void *stream,
                                    uint64_t *handle_out);
int32_t mi_vs_accumulate_end_pair(uint64_t ctx, uint64_t handle, mi_vs_partial *partials_out);
int32_t mi_vs_batches_per_pass(uint64_t ctx, size_t n);  /* >= 1, or a negative status */
int32_t mi_vs_batches_per_merged_pass(uint64_t ctx, size_t n);  /* the same for mi_vs_accumulate_begin_merged (below) */
int32_t mi_vs_finalize_async(uint64_t ctx, const mi_vs_partial *partials, size_t n_partials, uint64_t *ticket_out);
int32_t mi_vs_finalize_wait(uint64_t ctx, uint64_t ticket, uint8_t *all_ok);
/* Several pairing products under ONE ticket (one product stream, a workgroup per product): product k sums
 * partials[k * n_partials_each .. (k + 1) * n_partials_each).  What becomes ready together -- the batches of a lane
 * pass, a burst of chunks -- then needs one of the ten product slots, not one each.  _wait_multi writes all_ok[k];
 * `count` must be the ticket's own. */
#define MI_VS_MAX_PRODUCTS_PER_TICKET 4
int32_t mi_vs_finalize_async_multi(uint64_t ctx, const mi_vs_partial *partials, size_t n_partials_each, int32_t count,
                                   uint64_t *ticket_out);
int32_t mi_vs_finalize_wait_multi(uint64_t ctx, uint64_t ticket, uint8_t *all_ok, int32_t count);
/* After MI_ERR_AGAIN from mi_vs_finalize_async[_multi]: MI_OK as soon as a product slot of the device is free (somebody's
 * mi_vs_finalize_wait ended), MI_ERR_AGAIN if none was within timeout_ms.  For callers whose tickets in flight are not
 * the ones holding the slots (two drivers, or a driver and a caller of the split entry points, on one device). */
int32_t mi_vs_wait_product_slot(uint64_t ctx, int32_t timeout_ms);

/* ---- native streaming driver over the split entry points above (what rayon workers would otherwise hand-roll) ----
 * One driver owns host threads: `fold_threads` wait for accumulations and fold them, one finisher turns partials into
 * pairing products (up to `products` in flight) and hands verdicts back in submission order.  The caller's thread only
 * enqueues.  `exchange` (may be NULL) runs on the finisher thread, in batch order, once per GROUP of `exchange_group`
 * consecutive batches (1..8; the last group of a finished stream may be shorter; larger groups mean fewer collectives
 * and later verdicts): it receives this rank's `count` partials and must write all ranks' partials rank-major
 * (all[rank * count + k] = rank's partial of the group's k-th batch; multi-GPU: one all-gather of count * 592 bytes per
 * rank) and return the number of RANKS (world * count <= max_partials), or a negative value on failure. */
typedef int32_t (*mi_vs_exchange_fn)(void *user, const mi_vs_partial *mine, size_t count, mi_vs_partial *all, size_t max_partials);
int32_t mi_vs_stream_open(uint64_t ctx, int32_t fold_threads, int32_t products, int32_t lanes, mi_vs_exchange_fn exchange,
                          void *user, int32_t exchange_group, uint64_t *stream_out);
/* enqueue one batch (device pointers; blocks while every lane is taken); batches are numbered 0, 1, 2, ...
 * Under load (MI_SNARK_STREAM_PAIR_AT = 4 or more passes open) the batch may be HELD for the next submissions of the
 * same size on the same hip_stream to share its lane pass (MI_SNARK_STREAM_PER_PASS batches per pass: 2 with an
 * exchange, 4 without); a pass that ends, mi_vs_stream_finish, a blocking mi_vs_stream_next or a submission that cannot
 * join releases it.  WITHOUT an exchange the batches of a pass are folded into ONE aggregate
 * (mi_vs_accumulate_begin_merged; MI_SNARK_STREAM_MERGE=0 keeps them apart): one pairing product decides all of them
 * when it is one; when it is not, each of them is verified on its own (from its resident inputs) as its verdict is
 * taken -- every verdict is exactly what mi_vs_verify_batch_device would say.  The inputs must stay resident until the
 * batch's verdict has been taken (they always had to: a failed batch is taken apart from them).  One submitting thread
 * per stream; MI_ERR_AGAIN when two race on different hip_streams: retry. */
int32_t mi_vs_stream_submit(uint64_t stream, const void *d_proofs, size_t n, const void *d_randomizers128, void *hip_stream);
/* no more batches will be submitted: lets mi_vs_stream_next drain */
int32_t mi_vs_stream_finish(uint64_t stream);
/* next verdict in submission order.  block = 0: MI_ERR_AGAIN if it is not ready yet; after the last verdict of a
 * finished stream: MI_ERR_DONE.  A failure inside the pipeline is reported here (status of the failing call). */
int32_t mi_vs_stream_next(uint64_t stream, int32_t block, uint8_t *all_ok);
/* the same with the batch's verdict vector: *n_out = its size; verdicts[0..n) = 1 when all_ok, otherwise the invalid
 * proofs of THIS rank's shard are named by mi_vs_identify_device on the submitted buffers, which therefore must stay
 * valid until their verdict has been taken.  capacity < n: MI_ERR_BAD_ARGS (the verdict is consumed). */
int32_t mi_vs_stream_next_verdicts(uint64_t stream, int32_t block, uint8_t *all_ok, uint8_t *verdicts, size_t capacity,
                                   size_t *n_out);
int32_t mi_vs_stream_close(uint64_t stream); /* finishes, drains and joins the threads */

/* ---- the exchange itself, native: RCCL called from the library (librccl is dlopen'ed on first use) ----
 * One process per GPU: rank 0 calls mi_comm_unique_id, the 128 bytes reach the other ranks over the caller's own
 * transport, every rank calls mi_comm_init_rank (collective).  mi_vs_exchange_rccl is an mi_vs_exchange_fn: pass it to
 * mi_vs_stream_open with user = (void *)(uintptr_t)comm.  The transport is the LIBRARY's decision, taken by all ranks
 * together at mi_comm_init_rank: RCCL over xGMI when it comes up on every rank; otherwise (two ranks rehearsing on one
 * device, no librccl, an init that fails somewhere, MI_SNARK_EXCHANGE=host) a shared-memory all-gather between the
 * node's processes -- the ranks of SURVEY.md 8e share one node.  mi_comm_transport tells which.
 * One process, all GPUs: mi_comm_init_local (ncclCommInitAll over the library's device list; what mi_vs_verify_block
 * uses by itself). */
#define MI_COMM_ID_BYTES 128
int32_t mi_comm_unique_id(uint8_t *id_out /* MI_COMM_ID_BYTES */);
int32_t mi_comm_init_rank(const uint8_t *id, int32_t rank, int32_t world, uint64_t *comm_out);
int32_t mi_comm_init_local(uint64_t *comm_out);
int32_t mi_comm_world(uint64_t comm); /* ranks, or a negative status */
#define MI_COMM_RCCL 1
#define MI_COMM_HOST 2
int32_t mi_comm_transport(uint64_t comm); /* MI_COMM_RCCL / MI_COMM_HOST, or a negative status */
int32_t mi_comm_destroy(uint64_t comm);
/* all-gather of one `bytes`-sized host record per rank: all[rank * bytes ...] (rank-per-process communicators) */
int32_t mi_comm_all_gather(uint64_t comm, const void *mine, size_t bytes, void *all);
int32_t mi_vs_exchange_rccl(void *user, const mi_vs_partial *mine, size_t count, mi_vs_partial *all, size_t max_partials);

/* ---- compressed G1 points: snarkVM's CanonicalSerialize with compression (snarkvm-curves templates/
 * short_weierstrass_jacobian + SWFlags [RECALL: the crate is not in /root/reference]; replaces the per-point
 * `G1Affine::deserialize_compressed` a Rust caller runs on the commitments of a Proof / CircuitVerifyingKey found in
 * Transaction bytes: reference node/router/messages/src/unconfirmed_transaction.rs:25-28,
 * node/bft/events/src/block_response.rs:64-70).  48 bytes per point: x little-endian, below q; bit 7 of byte 47 =
 * y is the larger of the two roots (y > (q - 1) / 2); bit 6 = infinity (x = 0).  ok[i] = 0 (and out[i] = infinity)
 * for x >= q, a non-residue x^3 + 1, a malformed infinity and, with check_subgroup != 0, a curve point outside the
 * order-r subgroup (snarkVM checks that on deserialisation).  The _device form runs on `stream`; its input must be
 * 16-byte aligned. */
int32_t mi_g1_decompress(const uint8_t *compressed48, size_t n, int32_t check_subgroup, mi_g1_affine *out, uint8_t *ok);
int32_t mi_g1_decompress_device(const void *d_compressed48, size_t n, int32_t check_subgroup, void *d_out, void *d_ok, void *stream);
int32_t mi_g1_compress(const mi_g1_affine *points, size_t n, uint8_t *compressed48_out);

/* ---- batch-certificate signatures: Schnorr over Edwards-BLS12, many at once ----
 * Replaces the per-signature `signature.verify(&address, &[batch_id])` loop over a certificate's signers (reference
 * node/bft/src/helpers/storage.rs:465 check_certificate, node/bft/src/gateway.rs:1391 verify_bytes, account/src/lib.rs:76-88),
 * i.e. snarkVM's console::account::Signature::verify [RECALL: the crate is not in /root/reference -- parity unpinned]:
 *   g_r = response G + challenge pk_sig;  challenge == H8(g_r.x, pk_sig.x, pr_sig.x, address.x, message)  and
 *   address == pk_sig + pr_sig + H4(pk_sig.x, pr_sig.x) G
 * on the twisted Edwards curve -x^2 + y^2 = 1 + 3021 x^2 y^2 over Fr (subgroup order s of 251 bits, cofactor 4), H4 / H8 =
 * one permutation of a Poseidon sponge over Fr of rate 4 / 8 truncated to 250 bits.  The generator and the Poseidon
 * constants are this library's own (tools/gen_constants.py; snarkVM's are not available offline).  Coordinates and the
 * message are Montgomery Fr, the scalars canonical integers below s.  verdicts[i] = 1: valid.  Checked per record: fields
 * below r, scalars below s, the three points on the curve; NOT checked: membership of the points in the order-s subgroup
 * (snarkVM's deserialiser guarantees it for what it hands out).  Records here carry affine points; the 32-byte
 * x-coordinate encoding is a next step. */
typedef struct { mi_fr x, y; } mi_ed_affine; /* 64 B; the identity is (0, 1) */
typedef struct {
    mi_ed_affine address;
    mi_fr message;                 /* one field element: the batch id a certificate's signers sign */
    mi_bigint256 challenge, response;
    mi_ed_affine pk_sig, pr_sig;   /* the signature's compute key */
} mi_sig_record; /* 288 B */
int32_t mi_schnorr_verify_batch(const mi_sig_record *recs, size_t n, uint8_t *verdicts);
int32_t mi_schnorr_verify_batch_device(const void *d_recs, size_t n, void *d_verdicts, void *stream);

/* ---- crypto_hash::PoseidonSponge<Fq, 2, 1>, many independent sponges at once ----
 * Structure as snarkVM's (width 3, S-box x^17, 8 full + 31 partial rounds); the round constants and MDS matrix
 * are this library's own (tools/gen_constants.py): snarkVM's are not available offline, so this is the kernel a
 * real Fiat-Shamir transcript would run on, not a bit-compatible one.  Each of the n sponges absorbs m Fq elements
 * (two per permutation) and squeezes k.  inputs: n x m, outputs: n x k, Montgomery. */
int32_t mi_poseidon_fq_hash_many(const mi_fq *inputs, size_t n, size_t m, size_t k, mi_fq *outputs);
int32_t mi_poseidon_fq_hash_many_device(const void *d_inputs, size_t n, size_t m, size_t k, void *d_outputs, void *stream);

/* ---- diagnostics (parity tests of the device field / group arithmetic) ---- */
int32_t mi_dbg_fq_mul(const mi_fq *a, const mi_fq *b, mi_fq *out, size_t n);
int32_t mi_dbg_fr_mul(const mi_fr *a, const mi_fr *b, mi_fr *out, size_t n);
int32_t mi_dbg_g1_add(const mi_g1_affine *a, const mi_g1_affine *b, mi_g1_affine *out, size_t n);
/* the reduced-radix arithmetic of the MSM / transcript kernels (csrc/field28.hpp) through its conversions: out[i] = a[i] b[i]
 * R^-1 for even i, 4 a[i] b[i] R^-1 (via squarings and lazy additions) for odd i; out[i] = a[i] + b[i] through its group law */
int32_t mi_dbg_fq_mul28(const mi_fq *a, const mi_fq *b, mi_fq *out, size_t n);
int32_t mi_dbg_g1_add28(const mi_g1_affine *a, const mi_g1_affine *b, mi_g1_affine *out, size_t n);
int32_t mi_dbg_fq_mul_throughput(size_t n_threads, uint32_t iters, double *gmul_per_s);
/* the plan an MSM of n points with scalars of scalar_bits (254 / 128) runs on: window width c, windows, chunk length, and
 * whether it takes the latency-bound small path (work counts of the benchmark's ALU view) */
int32_t mi_dbg_msm_plan(size_t n, int32_t scalar_bits, int32_t *c, int32_t *nwin, int32_t *chunk, int32_t *small_path);
/* synthetic inputs made on the device: out[i] = k_i * G with k_i = splitmix64(seed + i) (distinct, in-subgroup) */
int32_t mi_dbg_gen_points_device(uint64_t seed, size_t n, void *d_out, void *stream);

/* ---- profiling: HIP-event timing of the dominant kernels on the stream they are launched on ----
 * mi_prof_enable(1) brackets the headline kernels (pairing, msm_accumulate, vs_derive, ntt_pass and the whole MSM
 * pipeline) with hipEvents, mi_prof_enable(2) every section, 0 switches it off; mi_prof_read sums the
 * elapsed time of all recorded launches of `kernel` ("msm_accumulate", "ntt_pass", "pairing", ...),
 * returns the launch count and the work units (points / elements) those launches processed, and
 * clears the records of that kernel. */
int32_t mi_prof_enable(int32_t on);
int32_t mi_prof_read(const char *kernel, double *total_ms, uint64_t *launches, uint64_t *units);

#ifdef __cplusplus
}
#endif
#endif
