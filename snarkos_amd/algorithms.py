"""Host-side mirror of the snarkvm_algorithms surface for the hot path, over the C ABI.

Same names and argument meaning as the Rust interface the reference reaches through
Ledger::check_next_block (reference node/bft/ledger-service/src/ledger.rs:345-347); SURVEY.md Appendix B:

    msm::VariableBase::msm(bases, scalars)                     -> VariableBase.msm
    fft::EvaluationDomain::{new, fft, ifft, coset_fft, ...}    -> EvaluationDomain
    polycommit::kzg10::KZG10::batch_check                      -> KZG10.batch_check
    snark::varuna::VarunaSNARK::verify_batch (VarunaShaped-v1) -> VarunaShaped.verify_batch

Arrays are numpy uint64 in the ABI layout (Montgomery limbs; MSM scalars canonical), or torch CUDA
tensors of dtype uint8/int64 holding the same bytes for the device-resident entry points.
Every call goes to the HIP library; there is no CPU implementation behind these classes.
"""
import ctypes as C
import os

import numpy as np

from . import lib as L


def _is_torch_cuda(x):
    return hasattr(x, "is_cuda") and x.is_cuda


class VariableBase:
    """snarkvm_algorithms::msm::VariableBase"""

    @staticmethod
    def msm(bases, scalars, stream=None):
        """sum_i scalars[i] * bases[i] over min(len(bases), len(scalars)) pairs.

        bases: (n, 12) uint64 Montgomery affine; scalars: (n, 4) uint64 canonical.  Returns the
        projective result as a (18,) uint64 Jacobian point, normalised to z = 1 (z = 0: infinity).
        """
        out = np.zeros(L.G1_JAC_U64, dtype=np.uint64)
        if _is_torch_cuda(bases):
            n = min(bases.numel() * bases.element_size() // 96, scalars.numel() * scalars.element_size() // 32)
            L.check(L.lib().mi_msm_g1_device(L.dptr(bases), L.dptr(scalars), C.c_size_t(n), L.ptr(out),
                                             L.stream_ptr(stream)), "mi_msm_g1_device")
            return out
        bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 12)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        n = min(len(bases), len(scalars))
        L.check(L.lib().mi_msm_g1(L.ptr(bases), L.ptr(scalars), C.c_size_t(n), L.ptr(out)), "mi_msm_g1")
        return out

    @staticmethod
    def sum_partials(partials):
        """Add per-rank MSM results ((k, 18) uint64, each as returned by msm): the combine step of a sharded MSM."""
        partials = np.ascontiguousarray(partials, dtype=np.uint64).reshape(-1, L.G1_JAC_U64)
        out = np.zeros(L.G1_JAC_U64, dtype=np.uint64)
        L.check(L.lib().mi_g1_sum(L.ptr(partials), C.c_size_t(len(partials)), L.ptr(out)), "mi_g1_sum")
        return out


class ResidentBases:
    """Bases kept in HBM across calls (SRS powers / verifying-key commitments)."""

    def __init__(self, bases):
        bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 12)
        self.n = len(bases)
        h = C.c_uint64(0)
        L.check(L.lib().mi_bases_register(L.ptr(bases), C.c_size_t(self.n), C.byref(h)), "mi_bases_register")
        self.handle = h.value

    def msm(self, scalars, offset=0):
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(L.G1_JAC_U64, dtype=np.uint64)
        L.check(L.lib().mi_msm_g1_h(C.c_uint64(self.handle), C.c_size_t(offset), L.ptr(scalars),
                                    C.c_size_t(len(scalars)), L.ptr(out)), "mi_msm_g1_h")
        return out

    def commit(self, coeffs, offset=0, stream=None):
        """kzg10::KZG10::commit (no hiding) with these bases as powers_of_beta_g: sum_i coeffs[i] * powers[offset+i].
        coeffs: (n, 4) uint64 Montgomery Fr (numpy, or a CUDA tensor to stay on the device)."""
        out = np.zeros(L.G1_JAC_U64, dtype=np.uint64)
        if _is_torch_cuda(coeffs):
            n = coeffs.numel() * coeffs.element_size() // 32
            L.check(L.lib().mi_kzg_commit_h_device(C.c_uint64(self.handle), C.c_size_t(offset), L.dptr(coeffs), C.c_size_t(n),
                                                   L.ptr(out), L.stream_ptr(stream)), "mi_kzg_commit_h_device")
            return out
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
        L.check(L.lib().mi_kzg_commit_h(C.c_uint64(self.handle), C.c_size_t(offset), L.ptr(coeffs), C.c_size_t(len(coeffs)),
                                        L.ptr(out)), "mi_kzg_commit_h")
        return out

    def open(self, coeffs, point):
        """kzg10::KZG10::open (no hiding) with these bases as powers: (p(point) as (4,) uint64 Montgomery, commitment to
        the witness polynomial (p(X) - p(point)) / (X - point) as a normalised Jacobian point)."""
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
        point = np.ascontiguousarray(point, dtype=np.uint64).reshape(4)
        value, wit = np.zeros(4, dtype=np.uint64), np.zeros(L.G1_JAC_U64, dtype=np.uint64)
        L.check(L.lib().mi_kzg_open_h(C.c_uint64(self.handle), L.ptr(coeffs), C.c_size_t(len(coeffs)), L.ptr(point),
                                      L.ptr(value), L.ptr(wit)), "mi_kzg_open_h")
        return value, wit

    def commit_lagrange(self, evals):
        """kzg10::KZG10::commit_lagrange with these bases as the Lagrange-basis SRS L_i(beta) G: the same MSM, over the
        polynomial's evaluations instead of its coefficients."""
        evals = np.ascontiguousarray(evals, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(L.G1_JAC_U64, dtype=np.uint64)
        L.check(L.lib().mi_kzg_commit_lagrange_h(C.c_uint64(self.handle), L.ptr(evals), C.c_size_t(len(evals)), L.ptr(out)),
                "mi_kzg_commit_lagrange_h")
        return out

    def commit_hiding(self, gamma_bases, coeffs, rand_coeffs):
        """KZG10::commit with a hiding polynomial: sum c_i powers[i] + sum rho_j gamma_powers[j] (`gamma_bases`: the
        ResidentBases of powers_of_beta_times_gamma_g)."""
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
        rho = np.ascontiguousarray(rand_coeffs, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(L.G1_JAC_U64, dtype=np.uint64)
        L.check(L.lib().mi_kzg_commit_hiding_h(C.c_uint64(self.handle), C.c_uint64(gamma_bases.handle), L.ptr(coeffs),
                                               C.c_size_t(len(coeffs)), L.ptr(rho), C.c_size_t(len(rho)), L.ptr(out)),
                "mi_kzg_commit_hiding_h")
        return out

    def open_hiding(self, gamma_bases, coeffs, rand_coeffs, point):
        """KZG10::open with hiding: (p(point), random_v = rho(point), witness commitment over both quotients)."""
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
        rho = np.ascontiguousarray(rand_coeffs, dtype=np.uint64).reshape(-1, 4)
        point = np.ascontiguousarray(point, dtype=np.uint64).reshape(4)
        value, rv, wit = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64), np.zeros(L.G1_JAC_U64, dtype=np.uint64)
        L.check(L.lib().mi_kzg_open_hiding_h(C.c_uint64(self.handle), C.c_uint64(gamma_bases.handle), L.ptr(coeffs),
                                             C.c_size_t(len(coeffs)), L.ptr(rho), C.c_size_t(len(rho)), L.ptr(point), L.ptr(value),
                                             L.ptr(rv), L.ptr(wit)), "mi_kzg_open_hiding_h")
        return value, rv, wit

    def release(self):
        if self.handle:
            L.check(L.lib().mi_bases_release(C.c_uint64(self.handle)), "mi_bases_release")
            self.handle = 0


class EvaluationDomain:
    """snarkvm_algorithms::fft::EvaluationDomain<Fr> (radix-2, size = next power of two)."""

    def __init__(self, num_coeffs):
        size = 1
        while size < num_coeffs:
            size <<= 1
        self.size = size
        self.log_size_of_group = size.bit_length() - 1
        if self.log_size_of_group > 47:
            raise ValueError("domain larger than the 2-adicity of Fr")

    @classmethod
    def new(cls, num_coeffs):
        return cls(num_coeffs)

    def _run(self, data, inverse, coset, stream=None):
        if _is_torch_cuda(data):
            assert data.numel() * data.element_size() == 32 * self.size
            L.check(L.lib().mi_ntt_fr_device(L.dptr(data), C.c_uint32(self.log_size_of_group), C.c_int32(inverse),
                                             C.c_int32(coset), L.stream_ptr(stream)), "mi_ntt_fr_device")
            return data
        a = np.asarray(data, dtype=np.uint64).reshape(-1, 4)
        if len(a) > self.size:
            raise ValueError("more coefficients than the domain holds")
        buf = np.zeros((self.size, 4), dtype=np.uint64)  # zero-padded to the domain size, as fft() does
        buf[:len(a)] = a
        L.check(L.lib().mi_ntt_fr(L.ptr(buf), C.c_uint32(self.log_size_of_group), C.c_int32(inverse), C.c_int32(coset)),
                "mi_ntt_fr")
        return buf

    def fft(self, coeffs, stream=None):
        return self._run(coeffs, 0, 0, stream)

    def ifft(self, evals, stream=None):
        return self._run(evals, 1, 0, stream)

    def coset_fft(self, coeffs, stream=None):
        return self._run(coeffs, 0, 1, stream)

    def coset_ifft(self, evals, stream=None):
        return self._run(evals, 1, 1, stream)

    def divide_by_vanishing_poly_on_coset(self, evals, vanishing_size, stream=None):
        """evals = the coset evaluations of a polynomial over THIS domain (coset_fft's output); divides them pointwise
        by Z_N(X) = X^N - 1, N = vanishing_size (a power of two <= the domain size).  coset_ifft of the result is the
        quotient.  In place for CUDA tensors."""
        log_n = int(vanishing_size).bit_length() - 1
        assert (1 << log_n) == vanishing_size and log_n <= self.log_size_of_group
        if _is_torch_cuda(evals):
            L.check(L.lib().mi_ntt_divide_by_vanishing_on_coset_device(L.dptr(evals), C.c_uint32(self.log_size_of_group),
                                                                       C.c_uint32(log_n), L.stream_ptr(stream)),
                    "mi_ntt_divide_by_vanishing_on_coset_device")
            return evals
        buf = np.ascontiguousarray(evals, dtype=np.uint64).reshape(self.size, 4).copy()
        L.check(L.lib().mi_ntt_divide_by_vanishing_on_coset(L.ptr(buf), C.c_uint32(self.log_size_of_group), C.c_uint32(log_n)),
                "mi_ntt_divide_by_vanishing_on_coset")
        return buf


class DensePolynomial:
    """snarkvm_algorithms::fft::DensePolynomial (the product only)"""

    @staticmethod
    def mul(a, b):
        """Coefficients of a * b, (len(a) + len(b) - 1, 4) uint64 Montgomery, through the evaluation domain."""
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
        b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros((len(a) + len(b) - 1, 4), dtype=np.uint64)
        L.check(L.lib().mi_poly_mul_fr(L.ptr(a), C.c_size_t(len(a)), L.ptr(b), C.c_size_t(len(b)), L.ptr(out)), "mi_poly_mul_fr")
        return out


class Bls12_377:
    """snarkvm_curves::bls12_377::Bls12_377 pairing engine"""

    @staticmethod
    def pairing(p, q):
        out = np.zeros(L.FQ12_U64, dtype=np.uint64)
        p = np.ascontiguousarray(p, dtype=np.uint64).reshape(12)
        q = np.ascontiguousarray(q, dtype=np.uint64).reshape(24)
        L.check(L.lib().mi_pairing(L.ptr(p), L.ptr(q), L.ptr(out)), "mi_pairing")
        return out

    @staticmethod
    def product_of_pairings_is_one(ps, qs):
        ps = np.ascontiguousarray(ps, dtype=np.uint64).reshape(-1, 12)
        qs = np.ascontiguousarray(qs, dtype=np.uint64).reshape(-1, 24)
        r = np.zeros(1, dtype=np.uint8)
        L.check(L.lib().mi_pairing_product_is_one(L.ptr(ps), L.ptr(qs), C.c_size_t(len(ps)), L.ptr(r)),
                "mi_pairing_product_is_one")
        return bool(r[0])


class PoseidonSponge:
    """snarkvm_algorithms::crypto_hash::PoseidonSponge<Fq, 2, 1>, many independent sponges per call
    (own round constants: see include/mi355x_snark.h)."""

    @staticmethod
    def hash_many(inputs, k, stream=None):
        """inputs: (n, m, 6) uint64 Montgomery Fq (or a CUDA tensor of the same bytes plus n, m via its shape);
        returns (n, k, 6)."""
        if _is_torch_cuda(inputs):
            import torch
            n, m = inputs.shape[0], inputs.shape[1]
            out = torch.empty((n, k, 6), dtype=inputs.dtype, device=inputs.device)
            L.check(L.lib().mi_poseidon_fq_hash_many_device(L.dptr(inputs), C.c_size_t(n), C.c_size_t(m), C.c_size_t(k),
                                                            L.dptr(out), L.stream_ptr(stream)), "mi_poseidon_fq_hash_many_device")
            return out
        a = np.ascontiguousarray(inputs, dtype=np.uint64)
        n, m = a.shape[0], a.shape[1]
        out = np.zeros((n, k, 6), dtype=np.uint64)
        L.check(L.lib().mi_poseidon_fq_hash_many(L.ptr(a), C.c_size_t(n), C.c_size_t(m), C.c_size_t(k), L.ptr(out)),
                "mi_poseidon_fq_hash_many")
        return out


class G1Codec:
    """`CanonicalDeserialize::deserialize_compressed` / `serialize_compressed` of `G1Affine`, a batch per call
    (48 bytes per point; convention in include/mi355x_snark.h, [RECALL])."""

    @staticmethod
    def decompress(data, check_subgroup=True, stream=None):
        """data: (n, 48) uint8 (numpy, or a 16-byte aligned CUDA tensor).  Returns (points (n, 12) uint64 Montgomery,
        ok (n,) uint8); a refused encoding yields ok = 0 and the point at infinity."""
        if _is_torch_cuda(data):
            import torch
            n = data.numel() // 48
            out = torch.empty((n, 12), dtype=torch.int64, device=data.device)
            ok = torch.empty(n, dtype=torch.uint8, device=data.device)
            L.check(L.lib().mi_g1_decompress_device(L.dptr(data), C.c_size_t(n), C.c_int32(1 if check_subgroup else 0), L.dptr(out),
                                                    L.dptr(ok), L.stream_ptr(stream)), "mi_g1_decompress_device")
            return out, ok
        d = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1, 48)
        out = np.zeros((len(d), 12), dtype=np.uint64)
        ok = np.zeros(len(d), dtype=np.uint8)
        L.check(L.lib().mi_g1_decompress(L.ptr(d), C.c_size_t(len(d)), C.c_int32(1 if check_subgroup else 0), L.ptr(out), L.ptr(ok)),
                "mi_g1_decompress")
        return out, ok

    @staticmethod
    def compress(points):
        pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 12)
        out = np.zeros((len(pts), 48), dtype=np.uint8)
        L.check(L.lib().mi_g1_compress(L.ptr(pts), C.c_size_t(len(pts)), L.ptr(out)), "mi_g1_compress")
        return out


class SchnorrBatch:
    """Batch-certificate signatures: Schnorr over Edwards-BLS12, a certificate's (or a subdag's)
    signatures verified in one call; the role of `signature.verify(&address, &[batch_id])` over the signers in
    Storage::check_certificate (reference node/bft/src/helpers/storage.rs:465)."""

    @staticmethod
    def verify(records, stream=None):
        """records: n x 288 bytes (mi_sig_record), host array or CUDA tensor -> uint8[n] verdicts (tensor in, tensor out)"""
        if _is_torch_cuda(records):
            import torch
            n = records.numel() * records.element_size() // L.SIG_RECORD_BYTES
            out = torch.empty(n, dtype=torch.uint8, device=records.device)
            L.check(L.lib().mi_schnorr_verify_batch_device(L.dptr(records), C.c_size_t(n), L.dptr(out), L.stream_ptr(stream)),
                    "mi_schnorr_verify_batch_device")
            return out
        r = np.ascontiguousarray(records).view(np.uint8).reshape(-1)
        n = r.size // L.SIG_RECORD_BYTES
        out = np.zeros(n, dtype=np.uint8)
        L.check(L.lib().mi_schnorr_verify_batch(L.ptr(r), C.c_size_t(n), L.ptr(out)), "mi_schnorr_verify_batch")
        return out


class KZG10:
    """snarkvm_algorithms::polycommit::kzg10::KZG10"""

    @staticmethod
    def batch_check(vk, commitments, points, values, proofs, randomizers128, gamma_g=None, random_v=None):
        """vk: 480-byte VerifierKey {g, h, beta_h}; proofs: the opening witnesses w_i.  With hiding: `gamma_g` (the
        VerifierKey's gamma_g, (12,) uint64) and `random_v` (KZGProof.random_v per opening)."""
        vk = np.ascontiguousarray(vk).view(np.uint8).reshape(-1)
        assert vk.size == L.KZG_VK_BYTES
        comms = np.ascontiguousarray(commitments, dtype=np.uint64).reshape(-1, 12)
        n = len(comms)
        pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(n, 4)
        vals = np.ascontiguousarray(values, dtype=np.uint64).reshape(n, 4)
        wit = np.ascontiguousarray(proofs, dtype=np.uint64).reshape(n, 12)
        rnd = np.ascontiguousarray(randomizers128, dtype=np.uint8).reshape(-1)
        assert rnd.size == 16 * n
        ok = np.zeros(1, dtype=np.uint8)
        if gamma_g is not None:
            gg = np.ascontiguousarray(gamma_g, dtype=np.uint64).reshape(12)
            rv = np.ascontiguousarray(random_v, dtype=np.uint64).reshape(n, 4)
            L.check(L.lib().mi_kzg_batch_check_hiding(L.ptr(vk), L.ptr(gg), L.ptr(comms), L.ptr(pts), L.ptr(vals), L.ptr(wit),
                                                      L.ptr(rv), C.c_size_t(n), L.ptr(rnd), L.ptr(ok)), "mi_kzg_batch_check_hiding")
            return bool(ok[0])
        L.check(L.lib().mi_kzg_batch_check(L.ptr(vk), L.ptr(comms), L.ptr(pts), L.ptr(vals), L.ptr(wit), C.c_size_t(n),
                                           L.ptr(rnd), L.ptr(ok)), "mi_kzg_batch_check")
        return bool(ok[0])

    @staticmethod
    def check(vk, commitment, point, value, proof):
        one = np.zeros(16, dtype=np.uint8)
        one[0] = 1
        return KZG10.batch_check(vk, [commitment], [point], [value], [proof], one)


class VarunaShaped:
    """Batch verifier for the synthetic VarunaShaped-v1 proofs; the role of
    VarunaSNARK::verify_batch inside VM::check_transactions."""

    def __init__(self, kzg_vk, circuit_vks, transcript="sha256", shape=None, key_ext=None, wire=False):
        """wire: the context takes WIRE records (MI_VS_WIRE: compressed points, canonical Fr), decoded on the device
        inside the lane pass; `wire_encode` turns uncompressed records into them.
        transcript: "sha256" (the round-1 stand-in) or "poseidon" (sponge over Fq with the reference's absorb volume,
        mi_vs_ctx_create_ex(..., MI_VS_FS_POSEIDON)); proofs are bound to their transcript kind.
        shape / key_ext: VarunaShaped-v2 (mi_vs_ctx_create_v2): the 56-byte mi_vs_shape (query-set table, degree-bound
        classes, hiding, linear-combination kinds, instances per proof) and the 480-byte mi_vs_key_ext (gamma_g, shift_h[2]);
        records are then 3184 bytes (mi_vs_proof2; 5504-byte mi_vs_proof3 when the shape carries instances)
        with three 128-bit randomizers (48 bytes) per proof."""
        vk = np.ascontiguousarray(kzg_vk).view(np.uint8).reshape(-1)
        cv = np.ascontiguousarray(circuit_vks).view(np.uint8).reshape(-1)
        assert vk.size == L.KZG_VK_BYTES and cv.size % L.VS_CIRCUIT_VK_BYTES == 0
        assert transcript in ("sha256", "poseidon")
        self.n_circuits = cv.size // L.VS_CIRCUIT_VK_BYTES
        h = C.c_uint64(0)
        flags = (L.VS_FS_POSEIDON if transcript == "poseidon" else 0) | (L.VS_WIRE if wire else 0)
        self.proof_bytes, self.rand_bytes = L.VS_PROOF_BYTES, 32
        if shape is not None:
            sh = np.ascontiguousarray(shape).view(np.uint8).reshape(-1)
            ext = None if key_ext is None else np.ascontiguousarray(key_ext).view(np.uint8).reshape(-1)
            assert sh.size == L.VS2_SHAPE_BYTES and (ext is None or ext.size == L.VS2_KEY_EXT_BYTES)
            L.check(L.lib().mi_vs_ctx_create_v2(L.ptr(vk), L.ptr(ext), L.ptr(sh), L.ptr(cv), C.c_size_t(self.n_circuits),
                                                C.c_uint32(flags), C.byref(h)), "mi_vs_ctx_create_v2")
            self.proof_bytes, self.rand_bytes = (L.VS3_PROOF_BYTES if int(sh[53]) > 1 else L.VS2_PROOF_BYTES), 48
            if int(sh[54]) > 1:
                self.proof_bytes = L.VS4_PROOF_BYTES
        else:
            L.check(L.lib().mi_vs_ctx_create_ex(L.ptr(vk), L.ptr(cv), C.c_size_t(self.n_circuits), C.c_uint32(flags), C.byref(h)),
                    "mi_vs_ctx_create_ex")
        self.ctx = h.value
        self.record_bytes = self.proof_bytes  # the uncompressed record of the context's kind
        pb, rb = C.c_size_t(0), C.c_size_t(0)
        L.check(L.lib().mi_vs_strides(C.c_uint64(self.ctx), C.byref(pb), C.byref(rb)), "mi_vs_strides")
        assert rb.value == self.rand_bytes and (wire or pb.value == self.proof_bytes)
        self.proof_bytes = pb.value  # what every entry point of this context reads per proof (the wire stride under MI_VS_WIRE)
        # accumulation lanes of this context (the library reads the same variable when it creates the context)
        try:
            lanes = int(os.environ.get("MI_SNARK_VS_LANES", L.VS_DEFAULT_LANES))
        except ValueError:
            lanes = L.VS_DEFAULT_LANES
        self.lanes = lanes if 1 <= lanes <= L.VS_MAX_OPEN_ACCUMULATIONS else L.VS_DEFAULT_LANES

    def wire_encode(self, records):
        """uncompressed records of this context's kind -> wire records (mi_vs_wire_encode, host arithmetic)"""
        r = np.ascontiguousarray(records).view(np.uint8).reshape(-1)
        n = r.size // self.record_bytes
        assert r.size == n * self.record_bytes
        wb = {L.VS_PROOF_BYTES: 1728, L.VS2_PROOF_BYTES: 2224, L.VS3_PROOF_BYTES: 4256, L.VS4_PROOF_BYTES: 5072}[self.record_bytes]
        out = np.zeros(n * wb, dtype=np.uint8)
        L.check(L.lib().mi_vs_wire_encode(C.c_uint64(self.ctx), L.ptr(r), C.c_size_t(n), L.ptr(out)), "mi_vs_wire_encode")
        return out

    def close(self):
        if self.ctx:
            L.check(L.lib().mi_vs_ctx_destroy(C.c_uint64(self.ctx)), "mi_vs_ctx_destroy")
            self.ctx = 0

    def verify_batch(self, proofs, randomizers128, stream=None):
        """Returns (all_ok: bool, verdicts: uint8[n])."""
        all_ok = np.zeros(1, dtype=np.uint8)
        if _is_torch_cuda(proofs):
            n = proofs.numel() * proofs.element_size() // self.proof_bytes
            verdicts = np.zeros(n, dtype=np.uint8)
            L.check(L.lib().mi_vs_verify_batch_device(C.c_uint64(self.ctx), L.dptr(proofs), C.c_size_t(n),
                                                      L.dptr(randomizers128), L.ptr(verdicts), L.ptr(all_ok),
                                                      L.stream_ptr(stream)), "mi_vs_verify_batch_device")
            return bool(all_ok[0]), verdicts
        p = np.ascontiguousarray(proofs).view(np.uint8).reshape(-1)
        n = p.size // self.proof_bytes
        r = np.ascontiguousarray(randomizers128, dtype=np.uint8).reshape(-1)
        assert r.size == self.rand_bytes * n
        verdicts = np.zeros(n, dtype=np.uint8)
        L.check(L.lib().mi_vs_verify_batch(C.c_uint64(self.ctx), L.ptr(p), C.c_size_t(n), L.ptr(r), L.ptr(verdicts),
                                           L.ptr(all_ok)), "mi_vs_verify_batch")
        return bool(all_ok[0]), verdicts

    def verify_block(self, proofs, randomizers128):
        """One block-validation batch over ALL devices of the process (mi_vs_verify_block): host arrays in, contiguous
        shards, one partial per device, all-gather (RCCL from the library), one pairing product.
        Returns (all_ok, verdicts[n])."""
        p = np.ascontiguousarray(proofs).view(np.uint8).reshape(-1)
        n = p.size // self.proof_bytes
        r = np.ascontiguousarray(randomizers128, dtype=np.uint8).reshape(-1)
        assert r.size == self.rand_bytes * n
        verdicts, all_ok = np.zeros(n, dtype=np.uint8), np.zeros(1, dtype=np.uint8)
        L.check(L.lib().mi_vs_verify_block(C.c_uint64(self.ctx), L.ptr(p), C.c_size_t(n), L.ptr(r), L.ptr(verdicts), L.ptr(all_ok)),
                "mi_vs_verify_block")
        return bool(all_ok[0]), verdicts

    def verify_block_device(self, shards):
        """The same with the shards already resident: `shards` = [(proofs_cuda, randomizers_cuda), ...], each pair on the
        device that is to verify it (mi_vs_verify_block_device).  Returns (all_ok, verdicts) in shard order."""
        k = len(shards)
        dp = (C.c_void_p * k)(*[L.dptr(p).value for p, _ in shards])
        dr = (C.c_void_p * k)(*[L.dptr(r).value for _, r in shards])
        cnt = (C.c_size_t * k)(*[p.numel() * p.element_size() // self.proof_bytes for p, _ in shards])
        n = sum(cnt)
        verdicts, all_ok = np.zeros(n, dtype=np.uint8), np.zeros(1, dtype=np.uint8)
        L.check(L.lib().mi_vs_verify_block_device(C.c_uint64(self.ctx), dp, dr, cnt, C.c_int32(k), L.ptr(verdicts), L.ptr(all_ok)),
                "mi_vs_verify_block_device")
        return bool(all_ok[0]), verdicts

    def accumulate(self, proofs_cuda, randomizers_cuda, stream=None):
        """One shard's accumulator (592 bytes): the unit exchanged between GPUs."""
        n = proofs_cuda.numel() * proofs_cuda.element_size() // self.proof_bytes
        part = np.zeros(L.VS_PARTIAL_BYTES, dtype=np.uint8)
        L.check(L.lib().mi_vs_accumulate_device(C.c_uint64(self.ctx), L.dptr(proofs_cuda), C.c_size_t(n),
                                                L.dptr(randomizers_cuda), L.ptr(part), L.stream_ptr(stream)),
                "mi_vs_accumulate_device")
        return part

    def accumulate_begin(self, proofs_cuda, randomizers_cuda, stream=None):
        """Enqueue one shard's accumulation (derive + both MSMs) without waiting; returns a handle."""
        n = proofs_cuda.numel() * proofs_cuda.element_size() // self.proof_bytes
        h = C.c_uint64(0)
        L.check(L.lib().mi_vs_accumulate_begin(C.c_uint64(self.ctx), L.dptr(proofs_cuda), C.c_size_t(n),
                                               L.dptr(randomizers_cuda), L.stream_ptr(stream), C.byref(h)),
                "mi_vs_accumulate_begin")
        return h.value

    def accumulate_end(self, handle):
        part = np.zeros(L.VS_PARTIAL_BYTES, dtype=np.uint8)
        L.check(L.lib().mi_vs_accumulate_end(C.c_uint64(self.ctx), C.c_uint64(handle), L.ptr(part)), "mi_vs_accumulate_end")
        return part

    def accumulate_begin_pair(self, proofs_a, randomizers_a, proofs_b, randomizers_b, stream=None):
        """Two batches in one lane pass (mi_vs_accumulate_begin_pair): each keeps its own partial.  Raises MiError
        (MI_ERR_BAD_ARGS) for shapes that do not pair (degree-bound classes, Poseidon transcript, too large)."""
        na = proofs_a.numel() * proofs_a.element_size() // self.proof_bytes
        nb = proofs_b.numel() * proofs_b.element_size() // self.proof_bytes
        h = C.c_uint64(0)
        L.check(L.lib().mi_vs_accumulate_begin_pair(C.c_uint64(self.ctx), L.dptr(proofs_a), C.c_size_t(na), L.dptr(randomizers_a),
                                                    L.dptr(proofs_b), C.c_size_t(nb), L.dptr(randomizers_b),
                                                    L.stream_ptr(stream), C.byref(h)), "mi_vs_accumulate_begin_pair")
        return h.value

    def accumulate_end_pair(self, handle):
        parts = np.zeros((2, L.VS_PARTIAL_BYTES), dtype=np.uint8)
        L.check(L.lib().mi_vs_accumulate_end_pair(C.c_uint64(self.ctx), C.c_uint64(handle), L.ptr(parts)), "mi_vs_accumulate_end_pair")
        return parts

    def accumulate_begin_multi(self, batches, stream=None):
        """Up to four (proofs, randomizers) batches in one lane pass (mi_vs_accumulate_begin_multi); returns a handle for
        accumulate_end_multi(handle, len(batches)).  Raises MiError (MI_ERR_BAD_ARGS) when they do not fit one pass."""
        k = len(batches)
        pr = (C.c_void_p * k)(*[L.dptr(p).value for p, _ in batches])
        rn = (C.c_void_p * k)(*[L.dptr(r).value for _, r in batches])
        ns = (C.c_size_t * k)(*[p.numel() * p.element_size() // self.proof_bytes for p, _ in batches])
        h = C.c_uint64(0)
        L.check(L.lib().mi_vs_accumulate_begin_multi(C.c_uint64(self.ctx), C.c_int32(k), pr, ns, rn, L.stream_ptr(stream), C.byref(h)),
                "mi_vs_accumulate_begin_multi")
        return h.value

    def accumulate_begin_merged(self, batches, stream=None):
        """Two to eight (proofs, randomizers) batches folded into ONE aggregate (mi_vs_accumulate_begin_merged); returns a
        handle for accumulate_end(handle): one partial over all their proofs.  Raises MiError (MI_ERR_BAD_ARGS) when they
        do not fit one pass."""
        k = len(batches)
        pr = (C.c_void_p * k)(*[L.dptr(p).value for p, _ in batches])
        rn = (C.c_void_p * k)(*[L.dptr(r).value for _, r in batches])
        ns = (C.c_size_t * k)(*[p.numel() * p.element_size() // self.proof_bytes for p, _ in batches])
        h = C.c_uint64(0)
        L.check(L.lib().mi_vs_accumulate_begin_merged(C.c_uint64(self.ctx), C.c_int32(k), pr, ns, rn, L.stream_ptr(stream), C.byref(h)),
                "mi_vs_accumulate_begin_merged")
        return h.value

    def accumulate_end_multi(self, handle, count):
        parts = np.zeros((count, L.VS_PARTIAL_BYTES), dtype=np.uint8)
        L.check(L.lib().mi_vs_accumulate_end_multi(C.c_uint64(self.ctx), C.c_uint64(handle), L.ptr(parts), C.c_int32(count)),
                "mi_vs_accumulate_end_multi")
        return parts

    def batches_per_pass(self, n_proofs):
        """How many batches of n_proofs one lane pass takes for this context's shape (1..4)."""
        r = L.lib().mi_vs_batches_per_pass(C.c_uint64(self.ctx), C.c_size_t(n_proofs))
        if r < 1:
            raise L.MiError(r, "mi_vs_batches_per_pass")
        return r

    def batches_per_merged_pass(self, n_proofs):
        """How many batches of n_proofs one MERGED pass takes (mi_vs_accumulate_begin_merged) for this shape (1..8)."""
        r = L.lib().mi_vs_batches_per_merged_pass(C.c_uint64(self.ctx), C.c_size_t(n_proofs))
        if r < 1:
            raise L.MiError(r, "mi_vs_batches_per_merged_pass")
        return r

    def finalize_multi(self, groups):
        """Several pairing products under one ticket (mi_vs_finalize_async_multi / _wait_multi): groups is a list of
        equally long lists of partials (one list per product: the shards of one batch); returns one bool per product."""
        k, each = len(groups), len(groups[0])
        assert all(len(g) == each for g in groups)
        parts = np.ascontiguousarray(np.stack([np.stack([np.asarray(p, dtype=np.uint8).reshape(-1) for p in g]) for g in groups]))
        t = C.c_uint64(0)
        L.check(L.lib().mi_vs_finalize_async_multi(C.c_uint64(self.ctx), L.ptr(parts), C.c_size_t(each), C.c_int32(k), C.byref(t)),
                "mi_vs_finalize_async_multi")
        ok = np.zeros(k, dtype=np.uint8)
        L.check(L.lib().mi_vs_finalize_wait_multi(C.c_uint64(self.ctx), t, L.ptr(ok), C.c_int32(k)), "mi_vs_finalize_wait_multi")
        return [bool(x) for x in ok]

    def finalize_async(self, partials):
        """Enqueue the pairing product for these partials on the context's own stream; returns a ticket."""
        parts = np.ascontiguousarray(partials, dtype=np.uint8).reshape(-1, L.VS_PARTIAL_BYTES)
        t = C.c_uint64(0)
        L.check(L.lib().mi_vs_finalize_async(C.c_uint64(self.ctx), L.ptr(parts), C.c_size_t(len(parts)), C.byref(t)),
                "mi_vs_finalize_async")
        return t.value

    def finalize_wait(self, ticket):
        ok = np.zeros(1, dtype=np.uint8)
        L.check(L.lib().mi_vs_finalize_wait(C.c_uint64(self.ctx), C.c_uint64(ticket), L.ptr(ok)), "mi_vs_finalize_wait")
        return bool(ok[0])

    def identify(self, proofs_cuda, randomizers_cuda, stream=None):
        """The verdict vector of a batch already known to fail (mi_vs_identify_device): what verify_batch would return,
        without its aggregate pass."""
        n = proofs_cuda.numel() * proofs_cuda.element_size() // self.proof_bytes
        verdicts = np.zeros(n, dtype=np.uint8)
        L.check(L.lib().mi_vs_identify_device(C.c_uint64(self.ctx), L.dptr(proofs_cuda), C.c_size_t(n), L.dptr(randomizers_cuda),
                                              L.ptr(verdicts), L.stream_ptr(stream)), "mi_vs_identify_device")
        return verdicts

    def verify_stream(self, batches, gather=None, depth=8, acc_depth=12, fold_threads=3, exchange_group=1, comm=None,
                      verdicts=False):
        """Pipelined verification of a stream of batches (chunks of a block, or mempool traffic): yields all_ok per
        batch, in order.

        A thin wrapper over the library's native driver (mi_vs_stream_*: csrc/stream.hip).  The caller's thread
        enqueues batch k (derive + both MSMs on one of `acc_depth` stream lanes); `fold_threads` native threads wait
        for accumulations and fold them on the host; one finisher thread takes the partials in batch order,
        exchanges them (`gather`, optional: maps this rank's (count, 592) partials of `exchange_group` consecutive
        batches to all ranks' (world, count, 592) -- one collective per group, issued from that one thread) and starts
        the pairing products; up to `depth` (<= 16, the device has MI_SNARK_VS_SLOTS = 8) product tickets are in flight, each on its own stream.  `batches` yields
        (proofs_cuda, randomizers_cuda).  `comm` (a handle of lib.comm_init_rank) makes the exchange native instead:
        the library's own mi_vs_exchange_rccl (ncclAllGather from the finisher thread, no Python on the data path).
        Submitted tensors are kept alive until their verdict has been yielded.
        verdicts=True yields (all_ok, verdict vector or None) instead (mi_vs_stream_next_verdicts): a failing batch is
        taken apart on the device from the still-held tensors; with an exchange, all_ok covers every rank's shard and
        each rank names the invalid proofs of its own.
        """
        lib = L.lib()
        keep = {}

        def exchange(_user, mine, count, out, max_parts):  # runs on the finisher thread (ctypes takes the GIL)
            try:
                part = np.ctypeslib.as_array(mine, shape=(count, L.VS_PARTIAL_BYTES)).copy()
                parts = np.ascontiguousarray(gather(part), dtype=np.uint8).reshape(-1, count, L.VS_PARTIAL_BYTES)
                world = len(parts)
                if world == 0 or world * count > max_parts:
                    return -1
                np.ctypeslib.as_array(out, shape=(world * count * L.VS_PARTIAL_BYTES,))[:] = parts.reshape(-1)
                return world
            except BaseException as e:  # surfaced by the generator below
                keep["error"] = e
                return L.MI_ERR_KERNEL

        user = None
        if comm is not None:
            cb, user = C.cast(lib.mi_vs_exchange_rccl, L.VS_EXCHANGE_FN), C.c_void_p(comm)
        elif gather is not None:
            cb = L.VS_EXCHANGE_FN(exchange)
        else:
            cb = C.cast(None, L.VS_EXCHANGE_FN)
        sid = C.c_uint64(0)
        L.check(lib.mi_vs_stream_open(C.c_uint64(self.ctx), C.c_int32(fold_threads), C.c_int32(depth),
                                      C.c_int32(max(1, min(acc_depth, self.lanes))), cb, user, C.c_int32(exchange_group), C.byref(sid)),
                "mi_vs_stream_open")
        ok = C.c_uint8(0)
        held = []  # submitted tensors, oldest first: the lanes read them until the batch's verdict exists

        def drain(block):
            while True:
                if verdicts:
                    cap = max([p.numel() * p.element_size() // self.proof_bytes for p, _ in held] or [1])
                    vec, n_out = np.zeros(cap, dtype=np.uint8), C.c_size_t(0)
                    st = lib.mi_vs_stream_next_verdicts(sid, 1 if block else 0, C.byref(ok), L.ptr(vec), C.c_size_t(cap), C.byref(n_out))
                else:
                    st = lib.mi_vs_stream_next(sid, 1 if block else 0, C.byref(ok))
                if st == L.MI_OK:
                    if held:
                        held.pop(0)
                    if not verdicts:
                        yield bool(ok.value)
                    else:
                        yield bool(ok.value), (None if ok.value else vec[:n_out.value].copy())
                elif st == L.MI_ERR_AGAIN or st == L.MI_ERR_DONE:
                    return
                else:
                    if "error" in keep:
                        raise keep["error"]
                    raise L.MiError(st, "mi_vs_stream_next")

        # the submitting loop is on the critical path of a short stream (every batch of a burst is submitted before the
        # first verdict exists): the caller's stream is looked up once, and a tensor submitted again (a bench, a retry)
        # keeps its pointer
        hip_stream = L.stream_ptr(None)
        submit, byref_ok = lib.mi_vs_stream_submit, C.byref(ok)
        last_p = last_r = None
        try:
            for proofs, rnd in batches:
                if proofs is not last_p:
                    last_p, dp = proofs, L.dptr(proofs)
                    n = C.c_size_t(proofs.numel() * proofs.element_size() // self.proof_bytes)
                if rnd is not last_r:
                    last_r, dr = rnd, L.dptr(rnd)
                held.append((proofs, rnd))
                st = submit(sid, dp, n, dr, hip_stream)
                if st != L.MI_OK:
                    if "error" in keep:
                        raise keep["error"]
                    raise L.MiError(st, "mi_vs_stream_submit")
                if verdicts:
                    yield from drain(False)
                else:  # the common case inline: poll without a generator frame per batch
                    while True:
                        st = lib.mi_vs_stream_next(sid, 0, byref_ok)
                        if st != L.MI_OK:
                            break
                        held.pop(0)
                        yield bool(ok.value)
                    if st != L.MI_ERR_AGAIN and st != L.MI_ERR_DONE:
                        if "error" in keep:
                            raise keep["error"]
                        raise L.MiError(st, "mi_vs_stream_next")
            L.check(lib.mi_vs_stream_finish(sid), "mi_vs_stream_finish")
            yield from drain(True)
        finally:
            lib.mi_vs_stream_close(sid)

    def finalize(self, partials):
        parts = np.ascontiguousarray(partials, dtype=np.uint8).reshape(-1, L.VS_PARTIAL_BYTES)
        ok = np.zeros(1, dtype=np.uint8)
        L.check(L.lib().mi_vs_finalize(C.c_uint64(self.ctx), L.ptr(parts), C.c_size_t(len(parts)), L.ptr(ok)), "mi_vs_finalize")
        return bool(ok[0])
