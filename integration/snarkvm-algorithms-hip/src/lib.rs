//! Safe wrappers over `libmi355x_snark.so` at the three gates where `snarkvm-algorithms` already switches to its
//! `cuda` feature, plus the batched verifier.  snarkOS itself does not change: it reaches this code
//! through `Ledger::check_next_block` / `check_transaction_basic`
//! (reference node/bft/ledger-service/src/ledger.rs:341,346; node/consensus/src/lib.rs:499).
//!
//! Contract (the cuda gate's): every wrapper returns `Err(status)` on a non-zero status and the caller falls back to
//! its CPU path; a status never encodes a verdict.
//!
//! Not compiled in the repository's image (no Rust toolchain there).  `ffi.rs` is generated from the C header.

pub mod ffi;

use core::ffi::c_void;
use ffi::*;
use snarkvm_curves::bls12_377::{Fq, Fr, G1Affine, G1Projective};
use snarkvm_fields::PrimeField;

pub type Status = i32;

fn check(st: i32) -> Result<(), Status> {
    if st == MI_OK { Ok(()) } else { Err(st) }
}

/// `mi_init(device_ids, n)`: the GPUs this process drives; `devices[0]` runs the host-pointer entry points.
pub fn init(devices: &[i32]) -> Result<(), Status> {
    check(unsafe { mi_init(devices.as_ptr(), devices.len() as i32) })
}

/// G1Affine is not `repr(C)` (x, y, infinity: bool), so pack explicitly -- never transmute.
fn pack(p: &G1Affine) -> mi_g1_affine {
    if p.infinity {
        return mi_g1_affine { x: mi_fq { l: [0; 6] }, y: mi_fq { l: [0; 6] } };
    }
    mi_g1_affine { x: mi_fq { l: p.x.0 .0 }, y: mi_fq { l: p.y.0 .0 } } // Montgomery limbs as held in memory
}

fn unpack(j: &mi_g1_jac) -> G1Projective {
    // outputs are normalised (z = 1, or z = 0 for infinity): Jacobian, as G1Projective is
    G1Projective::new(Fq::from_montgomery_limbs(j.x.l), Fq::from_montgomery_limbs(j.y.l), Fq::from_montgomery_limbs(j.z.l))
}

const ZERO_JAC: mi_g1_jac = mi_g1_jac { x: mi_fq { l: [0; 6] }, y: mi_fq { l: [0; 6] }, z: mi_fq { l: [0; 6] } };

/// Drop-in for the body of `VariableBase::msm` behind `#[cfg(feature = "hip")]` (G1, `scalars.len() > 1024`).
pub fn msm(bases: &[G1Affine], scalars: &[<Fr as PrimeField>::BigInteger]) -> Result<G1Projective, Status> {
    let n = bases.len().min(scalars.len());
    let packed: Vec<mi_g1_affine> = bases[..n].iter().map(pack).collect();
    let mut out = ZERO_JAC;
    // BigInteger256 is `[u64; 4]`, canonical: exactly mi_bigint256
    check(unsafe { mi_msm_g1(packed.as_ptr(), scalars.as_ptr() as *const mi_bigint256, n, &mut out) })?;
    Ok(unpack(&out))
}

/// SRS powers / verifying-key commitments kept resident in HBM on every device of the list.
pub struct ResidentBases(u64);

impl ResidentBases {
    pub fn new(bases: &[G1Affine]) -> Result<Self, Status> {
        let packed: Vec<mi_g1_affine> = bases.iter().map(pack).collect();
        let mut h = 0u64;
        check(unsafe { mi_bases_register(packed.as_ptr(), packed.len(), &mut h) })?;
        Ok(Self(h))
    }
    pub fn msm(&self, offset: usize, scalars: &[<Fr as PrimeField>::BigInteger]) -> Result<G1Projective, Status> {
        let mut out = ZERO_JAC;
        check(unsafe { mi_msm_g1_h(self.0, offset, scalars.as_ptr() as *const mi_bigint256, scalars.len(), &mut out) })?;
        Ok(unpack(&out))
    }
    /// `KZG10::commit` with `hiding_bound = None`: coefficients stay in Montgomery form (converted on the device).
    pub fn commit(&self, coeffs: &[Fr]) -> Result<G1Projective, Status> {
        let mut out = ZERO_JAC;
        check(unsafe { mi_kzg_commit_h(self.0, 0, coeffs.as_ptr() as *const mi_fr, coeffs.len(), &mut out) })?;
        Ok(unpack(&out))
    }
}

impl Drop for ResidentBases {
    fn drop(&mut self) {
        unsafe { mi_bases_release(self.0) };
    }
}

/// Drop-in for `EvaluationDomain::fft_helper_in_place` (natural order in and out; size >= 32, 32-byte field).
pub fn ntt(domain_log_size: u32, a: &mut [Fr], inverse: bool, coset: bool) -> Result<(), Status> {
    debug_assert_eq!(a.len(), 1usize << domain_log_size);
    check(unsafe { mi_ntt_fr(a.as_mut_ptr() as *mut mi_fr, domain_log_size, inverse as i32, coset as i32) })
}

/// `KZG10::batch_check`; `randomizers128`: 16 bytes per opening from the `rng` the reference threads down from
/// ledger.rs:346 (`&mut rand::thread_rng()`), never generated inside the library.
pub fn kzg_batch_check(vk: &mi_kzg_vk, comms: &[G1Affine], points: &[Fr], values: &[Fr], witnesses: &[G1Affine],
                       randomizers128: &[u8]) -> Result<bool, Status> {
    let n = comms.len();
    assert!(points.len() == n && values.len() == n && witnesses.len() == n && randomizers128.len() == 16 * n);
    let c: Vec<mi_g1_affine> = comms.iter().map(pack).collect();
    let w: Vec<mi_g1_affine> = witnesses.iter().map(pack).collect();
    let mut ok = 0u8;
    check(unsafe {
        mi_kzg_batch_check(vk, c.as_ptr(), points.as_ptr() as *const mi_fr, values.as_ptr() as *const mi_fr, w.as_ptr(), n,
                           randomizers128.as_ptr(), &mut ok)
    })?;
    Ok(ok != 0)
}

/// One verifier (KZG vk + circuit verifying keys), resident on every device that is handed work for it.
/// The record and randomizer sizes belong to the CONTEXT (v1 2256 / 32 bytes, v2 3184 / 48, with instances 5504, with
/// several circuits 6320; under `MI_VS_WIRE` the wire strides 1728 / 2224 / 4256 / 5072): they are read back from the
/// library once (`mi_vs_strides`) and every safe entry point takes BYTES and checks their length against them -- the
/// library copies `n * proof_bytes`, so a typed slice of the wrong record type would be an out-of-bounds read.
pub struct BatchVerifier {
    ctx: u64,
    proof_bytes: usize,
    rand_bytes: usize,
}

impl BatchVerifier {
    fn wrap(ctx: u64) -> Result<Self, Status> {
        let (mut pb, mut rb) = (0usize, 0usize);
        if let Err(st) = check(unsafe { mi_vs_strides(ctx, &mut pb, &mut rb) }) {
            unsafe { mi_vs_ctx_destroy(ctx) };
            return Err(st);
        }
        Ok(Self { ctx, proof_bytes: pb, rand_bytes: rb })
    }
    /// `wire`: the records are the node's own bytes (48-byte compressed commitments, canonical field elements:
    /// ledger.rs:326-331), decoded on the device.
    pub fn new(vk: &mi_kzg_vk, circuits: &[mi_vs_circuit_vk], poseidon: bool, wire: bool) -> Result<Self, Status> {
        let mut h = 0u64;
        let flags: u32 = (if poseidon { MI_VS_FS_POSEIDON as u32 } else { 0 }) | (if wire { MI_VS_WIRE as u32 } else { 0 });
        check(unsafe { mi_vs_ctx_create_ex(vk, circuits.as_ptr(), circuits.len(), flags, &mut h) })?;
        Self::wrap(h)
    }
    /// VarunaShaped-v2 / v3 (query-set table, hiding, degree-bound classes, instances, several circuits per proof,
    /// inverse-challenge coefficients): three 128-bit randomizers per proof.
    pub fn new_v2(vk: &mi_kzg_vk, ext: Option<&mi_vs_key_ext>, shape: &mi_vs_shape, circuits: &[mi_vs_circuit_vk], poseidon: bool,
                  wire: bool) -> Result<Self, Status> {
        let mut h = 0u64;
        let e = ext.map_or(std::ptr::null(), |x| x as *const mi_vs_key_ext);
        let flags: u32 = (if poseidon { MI_VS_FS_POSEIDON as u32 } else { 0 }) | (if wire { MI_VS_WIRE as u32 } else { 0 });
        check(unsafe { mi_vs_ctx_create_v2(vk, e, shape, circuits.as_ptr(), circuits.len(), flags, &mut h) })?;
        Self::wrap(h)
    }
    pub fn proof_bytes(&self) -> usize { self.proof_bytes }
    pub fn rand_bytes(&self) -> usize { self.rand_bytes }
    /// number of records in `proofs`, after checking both slices against the context's strides
    fn count(&self, proofs: &[u8], randomizers128: &[u8]) -> usize {
        assert!(self.proof_bytes > 0 && proofs.len() % self.proof_bytes == 0, "proofs: not a whole number of this context's records");
        let n = proofs.len() / self.proof_bytes;
        assert_eq!(randomizers128.len(), self.rand_bytes * n, "randomizers: rand_bytes per proof");
        n
    }
    /// The chunk of `VM::check_transactions` (<= 1000 executions): verdict per proof, on the primary device.
    pub fn verify_batch(&self, proofs: &[u8], randomizers128: &[u8]) -> Result<(bool, Vec<u8>), Status> {
        let n = self.count(proofs, randomizers128);
        let (mut verdicts, mut all_ok) = (vec![0u8; n], 0u8);
        check(unsafe { mi_vs_verify_batch(self.ctx, proofs.as_ptr() as *const mi_vs_proof, n, randomizers128.as_ptr(), verdicts.as_mut_ptr(), &mut all_ok) })?;
        Ok((all_ok != 0, verdicts))
    }
    /// One block over ALL devices of the process: sharded inside the library, partials exchanged over RCCL,
    /// one pairing product (what `Consensus::try_advance_to_next_block` needs: node/consensus/src/lib.rs:499).
    pub fn verify_block(&self, proofs: &[u8], randomizers128: &[u8]) -> Result<(bool, Vec<u8>), Status> {
        let n = self.count(proofs, randomizers128);
        let (mut verdicts, mut all_ok) = (vec![0u8; n], 0u8);
        check(unsafe { mi_vs_verify_block(self.ctx, proofs.as_ptr() as *const mi_vs_proof, n, randomizers128.as_ptr(), verdicts.as_mut_ptr(), &mut all_ok) })?;
        Ok((all_ok != 0, verdicts))
    }
    /// v1 records as a typed slice (contexts made by `new(.., wire = false)` only: checked)
    pub fn verify_batch_v1(&self, proofs: &[mi_vs_proof], randomizers128: &[u8]) -> Result<(bool, Vec<u8>), Status> {
        assert_eq!(self.proof_bytes, std::mem::size_of::<mi_vs_proof>(), "not a v1 context");
        let bytes = unsafe { std::slice::from_raw_parts(proofs.as_ptr() as *const u8, std::mem::size_of_val(proofs)) };
        self.verify_batch(bytes, randomizers128)
    }
    /// uncompressed records of this context's kind -> wire records (host arithmetic)
    pub fn wire_encode(&self, records: &[u8], record_bytes: usize, wire_bytes: usize) -> Result<Vec<u8>, Status> {
        assert!(record_bytes > 0 && records.len() % record_bytes == 0);
        let n = records.len() / record_bytes;
        let mut out = vec![0u8; n * wire_bytes];
        check(unsafe { mi_vs_wire_encode(self.ctx, records.as_ptr() as *const c_void, n, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// The verdict vector of a chunk already known to fail (a streamed chunk whose `all_ok` came back false), from
    /// its device-resident inputs: no aggregate pass.
    pub unsafe fn identify(&self, d_proofs: *const c_void, n: usize, d_randomizers128: *const c_void, hip_stream: *mut c_void)
                           -> Result<Vec<u8>, Status> {
        let mut verdicts = vec![0u8; n];
        check(mi_vs_identify_device(self.ctx, d_proofs, n, d_randomizers128, verdicts.as_mut_ptr(), hip_stream))?;
        Ok(verdicts)
    }
    /// Several chunks in ONE lane pass (at most `MI_VS_MAX_BATCHES_PER_PASS`; `batches_per_pass` tells how many of a
    /// size fit): each chunk keeps its own partial.  `chunks`: (device proofs, count, device randomizers).
    pub unsafe fn accumulate_multi(&self, chunks: &[(*const c_void, usize, *const c_void)], hip_stream: *mut c_void)
                                   -> Result<Vec<mi_vs_partial>, Status> {
        let k = chunks.len();
        let (pr, ns, rn): (Vec<_>, Vec<_>, Vec<_>) = (chunks.iter().map(|c| c.0).collect(), chunks.iter().map(|c| c.1).collect(),
                                                      chunks.iter().map(|c| c.2).collect());
        let mut h = 0u64;
        check(mi_vs_accumulate_begin_multi(self.ctx, k as i32, pr.as_ptr(), ns.as_ptr(), rn.as_ptr(), hip_stream, &mut h))?;
        let mut parts: Vec<mi_vs_partial> = Vec::with_capacity(k);
        check(mi_vs_accumulate_end_multi(self.ctx, h, parts.as_mut_ptr(), k as i32))?;
        parts.set_len(k);
        Ok(parts)
    }
    pub fn batches_per_pass(&self, n: usize) -> Result<usize, Status> {
        let r = unsafe { mi_vs_batches_per_pass(self.ctx, n) };
        if r < 1 { Err(r) } else { Ok(r as usize) }
    }
    /// Streamed chunks (mempool admission, block sync): `exchange` = None on one GPU, or
    /// `Some(ffi::mi_vs_exchange_rccl)` with `user` = a communicator of `mi_comm_init_rank` when one process per GPU
    /// verifies one block together.
    pub fn stream(&self, fold_threads: i32, products: i32, lanes: i32, exchange: mi_vs_exchange_fn, user: *mut c_void,
                  exchange_group: i32) -> Result<VerifyStream, Status> {
        let mut s = 0u64;
        check(unsafe { mi_vs_stream_open(self.ctx, fold_threads, products, lanes, exchange, user, exchange_group, &mut s) })?;
        Ok(VerifyStream(s))
    }
}

impl Drop for BatchVerifier {
    fn drop(&mut self) {
        unsafe { mi_vs_ctx_destroy(self.ctx) };
    }
}

pub struct VerifyStream(u64);

impl VerifyStream {
    /// Device pointers (the caller staged the chunk); the buffers must stay alive until the chunk's verdict is read.
    pub unsafe fn submit(&self, d_proofs: *const c_void, n: usize, d_randomizers128: *const c_void, hip_stream: *mut c_void) -> Result<(), Status> {
        check(mi_vs_stream_submit(self.0, d_proofs, n, d_randomizers128, hip_stream))
    }
    pub fn finish(&self) -> Result<(), Status> {
        check(unsafe { mi_vs_stream_finish(self.0) })
    }
    /// `Ok(Some(all_ok))`, `Ok(None)` when not ready (non-blocking) or drained.
    pub fn next(&self, block: bool) -> Result<Option<bool>, Status> {
        let mut ok = 0u8;
        match unsafe { mi_vs_stream_next(self.0, block as i32, &mut ok) } {
            MI_OK => Ok(Some(ok != 0)),
            MI_ERR_AGAIN | MI_ERR_DONE => Ok(None),
            st => Err(st),
        }
    }
    /// The same with the chunk's verdict vector (all ones when `all_ok`; otherwise the invalid proofs of this rank's
    /// shard named on the device from the submitted, still resident buffers).  `capacity` >= the largest chunk.
    pub fn next_verdicts(&self, block: bool, capacity: usize) -> Result<Option<(bool, Vec<u8>)>, Status> {
        let (mut ok, mut n) = (0u8, 0usize);
        let mut verdicts = vec![0u8; capacity];
        match unsafe { mi_vs_stream_next_verdicts(self.0, block as i32, &mut ok, verdicts.as_mut_ptr(), capacity, &mut n) } {
            MI_OK => {
                verdicts.truncate(n);
                Ok(Some((ok != 0, verdicts)))
            }
            MI_ERR_AGAIN | MI_ERR_DONE => Ok(None),
            st => Err(st),
        }
    }
}

impl Drop for VerifyStream {
    fn drop(&mut self) {
        unsafe { mi_vs_stream_close(self.0) };
    }
}
