// Host-side functions shared between the translation units of libmi355x_snark.so.
// Each .hip file owns its kernels; nothing here needs relocatable device code.
#pragma once
#include "curve.hpp"
#include "runtime.hpp"

namespace mi {

struct Fq12;
struct G2Affine;
struct G2Prepared;
struct PairingConsts;

// msm.hip -- d_points / d_scalars are device pointers; result normalised Jacobian (z = 1) or infinity (z = 0).
// scalar_bits bounds the scalars (254 = any canonical Fr, 128 = randomizers).
struct G1Xyzz;
struct MsmJob {
    size_t n = 0, groups = 0;  // groups: sums per problem
    int c = 0, nwin = 0;
    int problems = 1;  // up to 4 MSMs sharing one launch sequence (msm_enqueue's split points), h holds problems * groups sums
    bool per_bit = false;
    G1Xyzz *h = nullptr;  // pinned host copy of the window / bit sums
    hipEvent_t done = nullptr;  // recorded after the copy into h
};
// split points of a fused launch: points [0, at[0]) are problem 0, [at[0], at[1]) problem 1, ...; 0 = unused
#define MSM_MAX_PROBLEMS 8  // MSMs that may share one small-path launch sequence
struct MsmSplits {
    size_t at[MSM_MAX_PROBLEMS - 1] = {0, 0, 0, 0, 0, 0, 0};  // ascending; 0 = unused
    MsmSplits() {}
    MsmSplits(size_t a, size_t b = 0, size_t c = 0) : at{a, b, c, 0, 0, 0, 0} {}
    __host__ __device__ uint32_t problem_of(size_t i) const {
        uint32_t p = 0;
#pragma unroll
        for (int k = 0; k < MSM_MAX_PROBLEMS - 1; k++) p += (uint32_t)(i >= at[k]);
        return p;
    }
};
void msm_enqueue(Context &cx, const G1Affine *d_points, const uint32_t *d_scalars, size_t n, hipStream_t st, int scalar_bits,
                 int slot, MsmJob *job, int force_c = 0, int pin_parity = 0, MsmSplits split = MsmSplits());
// out_b .. out_d: the further problems' results when the job was enqueued with split points
void msm_finish(Context &cx, MsmJob *job, hipStream_t st, mi_g1_jac *out, mi_g1_jac *out_b = nullptr, mi_g1_jac *out_c = nullptr,
                mi_g1_jac *out_d = nullptr);
// all problems' results: outs[p] may be null (result not wanted), n_outs <= MSM_MAX_PROBLEMS
void msm_finish_all(Context &cx, MsmJob *job, hipStream_t st, mi_g1_jac *const *outs, int n_outs);
size_t msm_small_limit();  // largest n on the latency-bound small path (the only one that takes a split)
void msm_plan_query(size_t n, int scalar_bits, int *c, int *nwin, int *chunk, int *small);
void msm_device(Context &cx, const G1Affine *d_points, const uint32_t *d_scalars, size_t n, mi_g1_jac *out,
                hipStream_t st, int scalar_bits = 254, int force_c = 0);

// ntt.hip -- in place on d_data (2^log_n elements)
void ntt_device(Context &cx, Fr *d_data, uint32_t log_n, bool inverse, bool coset, hipStream_t st);
Fr host_omega(uint32_t log_n);

// pairing.hip.  verdict_only: the result is one exactly when the product is, and otherwise merely not one (the final
// exponentiation without its inversion in Fq: pairing_wg.hpp pw_final_exp); false: the value of the product
void pairing_product_prepared(Context &cx, const G1Affine *ps, const G2Prepared *const *d_qs, size_t n, Fq12 *out_host,
                              hipStream_t st, bool verdict_only);
void pairing_product_host_args(Context &cx, const G1Affine *ps, const G2Affine *qs, size_t n, Fq12 *out_host,
                               hipStream_t st, bool verdict_only);
void pairing_launch(Context &cx, const G1Affine *ps, const G2Prepared *const *d_qs, size_t n, Fq12 *d_out, hipStream_t st, int unit,
                    bool verdict_only);
void pairing_launch_multi(Context &cx, const G1Affine *ps, const G2Prepared *const *d_qs, size_t n, int count, Fq12 *d_out, hipStream_t st,
                          int unit);
const PairingConsts *pairing_consts(Context &cx, hipStream_t st);

// capi.hip: Poseidon over Fq (width 3, x^17, 4 + 31 + 4 rounds), constants resident per device
struct PoseidonConsts {
    Fq ark[MI_POSEIDON_ROUNDS][3];
    Fq mds[3][3];
};
const PoseidonConsts *poseidon_consts(Context &cx, hipStream_t st);

// verify.hip
int32_t kzg_batch_check_host(Context &cx, const mi_kzg_vk *vk, const mi_g1_affine *gamma_g, const mi_g1_affine *comms,
                             const mi_fr *points, const mi_fr *values, const mi_g1_affine *witnesses, const mi_fr *random_v,
                             size_t n, const uint8_t *rand128, uint8_t *ok_out);
int32_t vs_ctx_create(Context &cx, const mi_kzg_vk *vk, const mi_vs_key_ext *ext, const mi_vs_shape *shape,
                      const mi_vs_circuit_vk *circuits, size_t n_circuits, uint32_t flags, uint64_t *out);
// record and randomizer strides of a verifier context (v1: 2256 / 32, v2: 3184 / 48); false: no such context
bool vs_strides(uint64_t id, size_t *proof_bytes, size_t *rand_bytes);
int32_t vs_ctx_destroy(Context &cx, uint64_t id);
int32_t vs_wire_encode_host(uint64_t id, const void *records, size_t n, uint8_t *wire_out);
int32_t vs_accumulate_device(Context &cx, uint64_t id, const void *d_proofs, size_t n, const void *d_rand,
                             mi_vs_partial *out, hipStream_t st);
int32_t vs_accumulate_begin(Context &cx, uint64_t id, const void *d_proofs, size_t n, const void *d_rand, hipStream_t st,
                            uint64_t *handle);
int32_t vs_accumulate_end(Context &cx, uint64_t id, uint64_t handle, mi_vs_partial *out);
int32_t vs_accumulate_begin_multi(Context &cx, uint64_t id, int count, const void *const *d_proofs, const size_t *n,
                                  const void *const *d_rand, hipStream_t st, uint64_t *handle);
int32_t vs_accumulate_begin_merged(Context &cx, uint64_t id, int count, const void *const *d_proofs, const size_t *n,
                                   const void *const *d_rand, hipStream_t st, uint64_t *handle);
int32_t vs_accumulate_end_multi(Context &cx, uint64_t id, uint64_t handle, mi_vs_partial *outs, int count);
int32_t vs_batches_per_pass(Context &cx, uint64_t id, size_t n);
int32_t vs_batches_per_merged_pass(Context &cx, uint64_t id, size_t n);
int32_t vs_finalize(Context &cx, uint64_t id, const mi_vs_partial *parts, size_t np, uint8_t *all_ok, hipStream_t st);
int32_t vs_finalize_async(Context &cx, uint64_t id, const mi_vs_partial *parts, size_t np, uint64_t *ticket);
int32_t vs_finalize_wait(Context &cx, uint64_t id, uint64_t ticket, uint8_t *all_ok);
int32_t vs_finalize_async_multi(Context &cx, uint64_t id, const mi_vs_partial *parts, size_t np, int count, uint64_t *ticket);
int32_t vs_finalize_wait_multi(Context &cx, uint64_t id, uint64_t ticket, uint8_t *all_ok, int count);
int32_t vs_wait_product_slot(Context &cx, int timeout_ms);
int32_t vs_identify_device(Context &cx, uint64_t id, const void *d_proofs, size_t n, const void *d_rand, uint8_t *verdicts,
                           hipStream_t st);
int32_t vs_verify_batch_device(Context &cx, uint64_t id, const void *d_proofs, size_t n, const void *d_rand,
                               uint8_t *verdicts, uint8_t *all_ok, hipStream_t st);
void verify_release_keys();

// one block-validation batch sharded over the devices of the process (shard s on the device that owns d_proofs[s], or
// host arrays cut into contiguous shards over the whole device list); verdicts are laid out shard after shard
int32_t vs_verify_block(uint64_t id, const void *const *proofs, const void *const *rand, const size_t *n_per_shard,
                        int n_shards, bool device_ptrs, uint8_t *verdicts, uint8_t *all_ok);

// stream.hip / comm.hip
void stream_release_all();
void comm_release_all();
struct Comm;
std::shared_ptr<Comm> comm_local();  // RCCL communicator over the library's device list; nullptr: exchange on the host
int32_t comm_all_gather(Comm &c, const void *const *mine, size_t bytes, void *const *all);

}  // namespace mi
