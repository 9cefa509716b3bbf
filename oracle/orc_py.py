"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package snarkos_amd never imports this module.
All arrays are numpy uint64 with the ABI layout of include/mi355x_snark.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

VS_PROOF_BYTES = 2256
VS_CIRCUIT_VK_BYTES = 1184
KZG_VK_BYTES = 480


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("orc_core.c", "orc_algo.c", "orc_schnorr.c", "orc.h", "orc_internal.h", "mont_tmpl.h", "constants.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs if os.path.exists(s)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_self_test.restype = C.c_int
        _LIB.orc_ntt.restype = C.c_int
        for f in ("orc_kzg_check", "orc_kzg_batch_check", "orc_kzg_batch_check_hiding", "orc_vs2_verify_aggregated", "orc_vs3_verify_aggregated", "orc_vs_verify_one", "orc_vs_verify_aggregated",
                  "orc_g1_is_on_curve", "orc_g2_is_on_curve", "orc_fq12_is_one", "orc_pairing_product_is_one"):
            getattr(_LIB, f).restype = C.c_int
    return _LIB


def _p(a):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def u64(shape):
    return np.zeros(shape, dtype=np.uint64)


# ----------------------------------------------------------------------------- int <-> limb helpers
def int_to_limbs(v, n):
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)], dtype=np.uint64)


def limbs_to_int(a):
    return sum(int(x) << (64 * i) for i, x in enumerate(np.asarray(a).reshape(-1)))


def fr_from_ints(vals):
    """canonical python ints -> Montgomery (n,4) array"""
    can = np.stack([int_to_limbs(v, 4) for v in vals]) if len(vals) else u64((0, 4))
    out = u64(can.shape)
    for i in range(len(vals)):
        lib().orc_fr_from_canonical(_p(out[i]), _p(np.ascontiguousarray(can[i])))
    return out


def fr_to_ints(arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 4)
    out = []
    t = u64(4)
    for i in range(arr.shape[0]):
        lib().orc_fr_to_canonical(_p(t), _p(np.ascontiguousarray(arr[i])))
        out.append(limbs_to_int(t))
    return out


def fq_from_int(v):
    out = u64(6)
    lib().orc_fq_from_canonical(_p(out), _p(int_to_limbs(v, 6)))
    return out


def fq_to_int(a):
    t = u64(6)
    lib().orc_fq_to_canonical(_p(t), _p(np.ascontiguousarray(a, dtype=np.uint64)))
    return limbs_to_int(t)


def g1_from_ints(P):
    """affine python point (x, y) or None -> (12,) Montgomery array"""
    out = u64(12)
    if P is not None:
        out[:6] = fq_from_int(P[0])
        out[6:] = fq_from_int(P[1])
    return out


def g1_to_ints(a):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(12)
    if not a.any():
        return None
    return (fq_to_int(a[:6]), fq_to_int(a[6:]))


def g2_from_ints(Qp):
    out = u64(24)
    if Qp is not None:
        out[0:6] = fq_from_int(Qp[0][0])
        out[6:12] = fq_from_int(Qp[0][1])
        out[12:18] = fq_from_int(Qp[1][0])
        out[18:24] = fq_from_int(Qp[1][1])
    return out


def g2_to_ints(a):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(24)
    if not a.any():
        return None
    return ((fq_to_int(a[0:6]), fq_to_int(a[6:12])), (fq_to_int(a[12:18]), fq_to_int(a[18:24])))


def jac_to_affine(j):
    out = u64(12)
    lib().orc_g1_to_affine(_p(out), _p(np.ascontiguousarray(j, dtype=np.uint64)))
    return out


def fq12_to_power_basis(a):
    """(72,) tower array -> 12 python ints, coefficient of w^k at index k (k = i + 2j + 6b)."""
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(2, 3, 2, 6)
    out = [0] * 12
    for i in range(2):
        for j in range(3):
            for b in range(2):
                out[i + 2 * j + 6 * b] = fq_to_int(a[i, j, b])
    return out


# ----------------------------------------------------------------------------- algorithms
def g1_generator():
    g = u64(12)
    lib().orc_g1_generator(_p(g))
    return g


def g2_generator():
    g = u64(24)
    lib().orc_g2_generator(_p(g))
    return g


def g1_compress(points):
    """(n, 12) Montgomery affine points -> (n, 48) bytes (orc_g1_compress)"""
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 12)
    out = np.zeros((len(pts), 48), dtype=np.uint8)
    for i in range(len(pts)):
        lib().orc_g1_compress(_p(out[i]), _p(pts[i]))
    return out


def g1_decompress(data, check_subgroup=True):
    """(n, 48) bytes -> ((n, 12) Montgomery affine points, (n,) ok flags) (orc_g1_decompress)"""
    d = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1, 48)
    out = np.zeros((len(d), 12), dtype=np.uint64)
    ok = np.zeros(len(d), dtype=np.uint8)
    f = lib().orc_g1_decompress
    f.restype = C.c_int
    for i in range(len(d)):
        ok[i] = f(_p(out[i]), _p(d[i]), C.c_int(1 if check_subgroup else 0))
    return out, ok


def fq_sqrt(a):
    """canonical integer -> canonical square root, or None"""
    out = u64(6)
    f = lib().orc_fq_sqrt
    f.restype = C.c_int
    return fq_to_int(out) if f(_p(out), _p(fq_from_int(a))) else None


def g1_mul(p, k):
    out = u64(18)
    lib().orc_g1_mul(_p(out), _p(np.ascontiguousarray(p)), _p(int_to_limbs(k, 4)))
    return jac_to_affine(out)


def g2_mul(p, k):
    out = u64(24)
    lib().orc_g2_mul(_p(out), _p(np.ascontiguousarray(p)), _p(int_to_limbs(k, 4)))
    return out


def fixed_base_batch(k_mont):
    k_mont = np.ascontiguousarray(k_mont, dtype=np.uint64).reshape(-1, 4)
    out = u64((k_mont.shape[0], 12))
    lib().orc_g1_fixed_base_batch(_p(out), _p(k_mont), C.c_size_t(k_mont.shape[0]))
    return out


def msm_naive(bases, scalars):
    n = min(len(bases), len(scalars))
    out = u64(18)
    lib().orc_msm_naive(_p(out), _p(np.ascontiguousarray(bases)), _p(np.ascontiguousarray(scalars)), C.c_size_t(n))
    return jac_to_affine(out)


def msm_pippenger(bases, scalars, threads=1):
    n = min(len(bases), len(scalars))
    out = u64(18)
    lib().orc_msm_pippenger(_p(out), _p(np.ascontiguousarray(bases)), _p(np.ascontiguousarray(scalars)),
                            C.c_size_t(n), C.c_int(threads))
    return jac_to_affine(out)


def ntt(data, inverse=False, coset=False, threads=1):
    a = np.array(data, dtype=np.uint64, copy=True).reshape(-1, 4)
    log_n = a.shape[0].bit_length() - 1
    assert 1 << log_n == a.shape[0]
    rc = lib().orc_ntt(_p(a), C.c_uint32(log_n), C.c_int(int(inverse)), C.c_int(int(coset)), C.c_int(threads))
    assert rc == 0
    return a


def dft_naive(data):
    a = np.ascontiguousarray(data, dtype=np.uint64).reshape(-1, 4)
    out = u64(a.shape)
    lib().orc_dft_naive(_p(out), _p(a), C.c_uint32(a.shape[0].bit_length() - 1))
    return out


def pairing(p, q):
    out = u64(72)
    lib().orc_pairing(_p(out), _p(np.ascontiguousarray(p)), _p(np.ascontiguousarray(q)))
    return out


def kzg_vk(tau):
    vk = u64(KZG_VK_BYTES // 8)
    vk[0:12] = g1_generator()
    vk[12:36] = g2_generator()
    vk[36:60] = g2_mul(g2_generator(), tau)
    return vk


def kzg_batch_check(vk, comms, points, values, witnesses, randomizers):
    n = len(comms)
    rnd = np.ascontiguousarray(randomizers, dtype=np.uint8).reshape(-1)
    assert rnd.size == 16 * n
    return bool(lib().orc_kzg_batch_check(_p(vk), _p(np.ascontiguousarray(comms)), _p(np.ascontiguousarray(points)),
                                          _p(np.ascontiguousarray(values)), _p(np.ascontiguousarray(witnesses)),
                                          C.c_size_t(n), _p(rnd)))


def kzg_batch_check_hiding(vk, gamma_g, comms, points, values, witnesses, random_v, randomizers):
    n = len(comms)
    rnd = np.ascontiguousarray(randomizers, dtype=np.uint8).reshape(-1)
    assert rnd.size == 16 * n
    return bool(lib().orc_kzg_batch_check_hiding(_p(vk), _p(np.ascontiguousarray(gamma_g)), _p(np.ascontiguousarray(comms)),
                                                 _p(np.ascontiguousarray(points)), _p(np.ascontiguousarray(values)),
                                                 _p(np.ascontiguousarray(witnesses)), _p(np.ascontiguousarray(random_v)),
                                                 C.c_size_t(n), _p(rnd)))


def kzg_commit(powers, coeffs, threads=1):
    """sum_i coeffs[i] * powers[i] (coefficients Montgomery) -> (12,) affine"""
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
    out = u64(18)
    lib().orc_kzg_commit(_p(out), _p(np.ascontiguousarray(powers, dtype=np.uint64)), _p(coeffs), C.c_size_t(len(coeffs)), C.c_int(threads))
    return jac_to_affine(out)


def poly_mul_schoolbook(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
    out = u64((len(a) + len(b) - 1, 4))
    lib().orc_poly_mul_schoolbook(_p(out), _p(a), C.c_size_t(len(a)), _p(b), C.c_size_t(len(b)))
    return out


def poly_eval(coeffs, z):
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
    out = u64(4)
    lib().orc_poly_eval(_p(out), _p(coeffs), C.c_size_t(len(coeffs)), _p(np.ascontiguousarray(z, dtype=np.uint64)))
    return out


def poly_witness(coeffs, z):
    """(q, p(z)) with p(x) - p(z) = (x - z) q(x)"""
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
    q, v = u64((max(len(coeffs) - 1, 1), 4)), u64(4)
    lib().orc_poly_witness(_p(q), _p(v), _p(coeffs), C.c_size_t(len(coeffs)), _p(np.ascontiguousarray(z, dtype=np.uint64)))
    return q[:len(coeffs) - 1], v


def poseidon_hash_many(inputs, k, threads=1):
    """inputs: (n, m, 6) Montgomery Fq -> (n, k, 6)"""
    a = np.ascontiguousarray(inputs, dtype=np.uint64)
    n, m = a.shape[0], a.shape[1]
    out = u64((n, k, 6))
    lib().orc_poseidon_hash_many(_p(a), C.c_size_t(n), C.c_size_t(m), C.c_size_t(k), _p(out), C.c_int(threads))
    return out


def sha256(b):
    out = np.zeros(32, dtype=np.uint8)
    buf = np.frombuffer(bytes(b), dtype=np.uint8).copy() if len(b) else np.zeros(1, dtype=np.uint8)
    lib().orc_sha256(_p(out), _p(buf), C.c_size_t(len(b)))
    return out.tobytes()


FS_SHA256, FS_POSEIDON = 0, 1


class VSBatch:
    """A synthetic VarunaShaped-v1 batch in ABI layout.  `transcript`: FS_SHA256 or FS_POSEIDON (the oracle's
    process-wide switch is set before every call that derives challenges)."""

    transcript = FS_SHA256

    def _fs(self):
        lib().orc_vs_set_transcript(C.c_int(self.transcript))

    def __init__(self, seed, n, n_circuits=8, corrupt_per_1024=0, transcript=FS_SHA256):
        self.n, self.n_circuits, self.seed = n, n_circuits, seed
        self.transcript = transcript
        self._fs()
        self.vk = np.zeros(KZG_VK_BYTES, dtype=np.uint8)
        self.circuits = np.zeros(n_circuits * VS_CIRCUIT_VK_BYTES, dtype=np.uint8)
        self.proofs = np.zeros(n * VS_PROOF_BYTES, dtype=np.uint8)
        self.randomizers = np.zeros(n * 32, dtype=np.uint8)
        self.expected = np.zeros(n, dtype=np.uint8)
        lib().orc_vs_generate(C.c_uint64(seed), C.c_size_t(n), C.c_size_t(n_circuits), C.c_uint32(corrupt_per_1024),
                              _p(self.vk), _p(self.circuits), _p(self.proofs), _p(self.randomizers), _p(self.expected))

    def verify_cpu(self, threads=1):
        self._fs()
        v = np.zeros(self.n, dtype=np.uint8)
        lib().orc_vs_verify_batch_cpu(_p(self.vk), _p(self.circuits), _p(self.proofs), C.c_size_t(self.n),
                                      _p(self.randomizers), _p(v), C.c_int(threads))
        return v

    def verify_aggregated(self, threads=1):
        self._fs()
        alg = np.zeros(self.n, dtype=np.uint8)
        tc, tw = u64(12), u64(12)
        ok = lib().orc_vs_verify_aggregated(_p(self.vk), _p(self.circuits), C.c_size_t(self.n_circuits), _p(self.proofs),
                                            C.c_size_t(self.n), _p(self.randomizers), _p(alg), _p(tc), _p(tw), C.c_int(threads))
        return bool(ok), alg, tc, tw

    def challenges(self, i):
        self._fs()
        p = self.proofs[i * VS_PROOF_BYTES:(i + 1) * VS_PROOF_BYTES]
        circuit = int(np.frombuffer(p[:4].tobytes(), dtype=np.uint32)[0])
        cvk = self.circuits[circuit * VS_CIRCUIT_VK_BYTES:(circuit + 1) * VS_CIRCUIT_VK_BYTES]
        zb, zg, xi = u64(4), u64(4), u64(4)
        lib().orc_vs_challenges(_p(np.ascontiguousarray(cvk)), _p(np.ascontiguousarray(p)), _p(zb), _p(zg), _p(xi))
        return fr_to_ints(zb)[0], fr_to_ints(zg)[0], fr_to_ints(xi)[0]


VS2_PROOF_BYTES, VS2_KEY_EXT_BYTES, VS2_SHAPE_BYTES, VS2_RAND_BYTES = 3184, 480, 56, 48
VS1_QUERY_OF = [0] * 5 + [1] * 4 + [0] * 6 + [1] * 6


VS3_PROOF_BYTES, VS_MAX_INSTANCES = 5504, 4


VS4_PROOF_BYTES, VS_MAX_CIRCUITS = 6320, 3
VS_WIRE_BYTES = {1: 1728, 2: 2224, 3: 4256, 4: 5072}


def vs2_record_kind(shape):
    """1: mi_vs_proof (no shape), 2: mi_vs_proof2, 3: mi_vs_proof3 (instances), 4: mi_vs_proof4 (several circuits)"""
    if shape is None:
        return 1
    sh = np.asarray(shape).reshape(-1)
    return 4 if int(sh[54]) > 1 else (3 if int(sh[53]) > 1 else 2)


def vs2_record_bytes(shape):
    """bytes per proof record under a shape: mi_vs_proof2, mi_vs_proof3 when the shape carries instances, mi_vs_proof4
    when it carries several circuits"""
    return {1: VS_PROOF_BYTES, 2: VS2_PROOF_BYTES, 3: VS3_PROOF_BYTES, 4: VS4_PROOF_BYTES}[vs2_record_kind(shape)]


def vs_wire_bytes(shape):
    return VS_WIRE_BYTES[vs2_record_kind(shape)]


def vs_wire_encode(shape, records):
    """uncompressed records -> wire records (points compressed, Fr canonical); shape None: v1 records"""
    records = np.ascontiguousarray(records, dtype=np.uint8).reshape(-1)
    n = records.size // vs2_record_bytes(shape)
    out = np.zeros(n * vs_wire_bytes(shape), dtype=np.uint8)
    lib().orc_vs_wire_encode(_p(shape) if shape is not None else None, _p(records), C.c_size_t(n), _p(out))
    return out


def vs_wire_decode(shape, wire, threads=1):
    """wire records -> (uncompressed records, dec_ok[n])"""
    wire = np.ascontiguousarray(wire, dtype=np.uint8).reshape(-1)
    n = wire.size // vs_wire_bytes(shape)
    out = np.zeros(n * vs2_record_bytes(shape), dtype=np.uint8)
    ok = np.zeros(n, dtype=np.uint8)
    lib().orc_vs_wire_decode(_p(shape) if shape is not None else None, _p(wire), C.c_size_t(n), _p(out), _p(ok), C.c_int(threads))
    return out, ok


def vs2_shape(query_of=None, bound_class=None, hiding=True, n_bounds=2, lc_kind=None, max_instances=1, max_circuits=1):
    """56-byte mi_vs_shape / orc_vs_shape: query table, degree-bound classes, hiding flag, linear-combination kinds
    (0: xi^p; bit 0: times z_beta^16 - 1; bit 1: times z_gamma^16 - 1; bit 2: times 1 / (z_beta - z_gamma)), instances
    per proof, distinct circuits per proof"""
    q = list(query_of if query_of is not None else VS1_QUERY_OF)
    bc = list(bound_class if bound_class is not None else [-1, -1, -1, -1, -1, 0, 0, 1, 1])
    lc = list(lc_kind if lc_kind is not None else [0] * 21)
    assert len(q) == 21 and len(bc) == 9 and q[0] == 0 and q[1] == 0, "p_0 and p_1 open at the beta query (public-input identity)"
    assert len(lc) == 21 and lc[0] == 0 and lc[1] == 0 and all(0 <= x <= 7 for x in lc)
    out = np.zeros(VS2_SHAPE_BYTES, dtype=np.uint8)
    out[:21] = q
    out[21:30] = np.array(bc, dtype=np.int8).view(np.uint8)
    out[30], out[31] = int(bool(hiding)), n_bounds
    out[32:53] = lc
    assert 1 <= max_instances <= VS_MAX_INSTANCES
    out[53] = max_instances
    assert 1 <= max_circuits <= VS_MAX_CIRCUITS and max_circuits <= max_instances
    out[54] = max_circuits
    return out


class VS2Batch:
    """A synthetic VarunaShaped-v2 batch in ABI layout (query-set table, hiding, degree-bound classes)."""

    def __init__(self, seed, n, n_circuits=8, corrupt_per_1024=0, shape=None, transcript=FS_SHA256):
        self.n, self.n_circuits, self.seed, self.transcript = n, n_circuits, seed, transcript
        self.shape = vs2_shape() if shape is None else np.ascontiguousarray(shape, dtype=np.uint8)
        self.vk = np.zeros(KZG_VK_BYTES, dtype=np.uint8)
        self.ext = np.zeros(VS2_KEY_EXT_BYTES, dtype=np.uint8)
        self.circuits = np.zeros(n_circuits * VS_CIRCUIT_VK_BYTES, dtype=np.uint8)
        self.record_bytes = vs2_record_bytes(self.shape)
        self.proofs = np.zeros(n * self.record_bytes, dtype=np.uint8)
        self.randomizers = np.zeros(n * VS2_RAND_BYTES, dtype=np.uint8)
        self.expected = np.zeros(n, dtype=np.uint8)
        self.dec_ok = None
        lib().orc_vs_set_transcript(C.c_int(transcript))
        lib().orc_vs4_generate(C.c_uint64(seed), C.c_size_t(n), C.c_size_t(n_circuits), C.c_uint32(corrupt_per_1024), _p(self.shape),
                               _p(self.vk), _p(self.ext), _p(self.circuits), _p(self.proofs), _p(self.randomizers), _p(self.expected))

    @classmethod
    def wrap(cls, vk, ext, shape, circuits, proofs, randomizers, transcript=FS_SHA256):
        b = cls.__new__(cls)
        b.vk, b.ext, b.shape = (np.ascontiguousarray(x, dtype=np.uint8).reshape(-1) for x in (vk, ext, shape))
        b.circuits = np.ascontiguousarray(circuits, dtype=np.uint8).reshape(-1)
        b.proofs = np.ascontiguousarray(proofs, dtype=np.uint8).reshape(-1)
        b.randomizers = np.ascontiguousarray(randomizers, dtype=np.uint8).reshape(-1)
        b.record_bytes = vs2_record_bytes(b.shape)
        b.n, b.n_circuits = b.proofs.size // b.record_bytes, b.circuits.size // VS_CIRCUIT_VK_BYTES
        b.transcript, b.seed, b.expected, b.dec_ok = transcript, None, None, None
        return b

    @classmethod
    def from_wire(cls, vk, ext, shape, circuits, wire, randomizers, transcript=FS_SHA256, threads=1):
        """the batch a verifier sees when the records arrive in the wire form: decoded (orc_vs_wire_decode), with the
        per-proof decode flags that every verdict below takes into account"""
        recs, ok = vs_wire_decode(shape, wire, threads)
        b = cls.wrap(vk, ext, shape, circuits, recs, randomizers, transcript)
        b.dec_ok = ok
        return b

    def wire(self):
        return vs_wire_encode(self.shape, self.proofs)

    def verify_cpu(self, threads=1):
        lib().orc_vs_set_transcript(C.c_int(self.transcript))
        v = np.zeros(self.n, dtype=np.uint8)
        lib().orc_vs4_verify_batch_cpu(_p(self.vk), _p(self.ext), _p(self.shape), _p(self.circuits), C.c_size_t(self.n_circuits),
                                       _p(self.proofs), C.c_size_t(self.n), _p(self.randomizers),
                                       _p(self.dec_ok) if self.dec_ok is not None else None, _p(v), C.c_int(threads))
        return v

    def verify_aggregated(self, threads=1):
        """-> (all_ok, alg_ok[n], totals (4, 12): total_c, total_w, bound[0], bound[1] affine, zeros = infinity)"""
        lib().orc_vs_set_transcript(C.c_int(self.transcript))
        alg = np.zeros(self.n, dtype=np.uint8)
        tot = u64((4, 12))
        ok = lib().orc_vs4_verify_aggregated(_p(self.vk), _p(self.ext), _p(self.shape), _p(self.circuits), C.c_size_t(self.n_circuits),
                                             _p(self.proofs), C.c_size_t(self.n), _p(self.randomizers),
                                             _p(self.dec_ok) if self.dec_ok is not None else None, _p(alg), _p(tot), C.c_int(threads))
        return bool(ok), alg, tot


def g1_add_affine(a, b):
    ja, j = u64(18), u64(18)
    lib().orc_g1_from_affine(_p(ja), _p(np.ascontiguousarray(a, dtype=np.uint64)))
    lib().orc_g1_add_mixed(_p(j), _p(ja), _p(np.ascontiguousarray(b, dtype=np.uint64)))
    return jac_to_affine(j)


PARTIAL_BYTES = 592
_ONE6 = None


def _partial_pack(points, n, n_alg_fail):
    """(k, 12) affine points ((0,0) = infinity) -> the library's mi_vs_partial: four normalised Jacobian points
    (total_c, total_w, bound[0], bound[1]; missing ones are infinity) and the two counts"""
    global _ONE6
    if _ONE6 is None:
        _ONE6 = int_to_limbs((1 << 384) % 0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001, 6)
    part = np.zeros(74, dtype=np.uint64)
    for k in range(4):
        pt = points[k] if k < len(points) else np.zeros(12, dtype=np.uint64)
        if pt.any():
            part[18 * k:18 * k + 12] = pt
            part[18 * k + 12:18 * k + 18] = _ONE6
        else:
            part[18 * k:18 * k + 6] = _ONE6
            part[18 * k + 6:18 * k + 12] = _ONE6
    part[72] = n
    part[73] = n_alg_fail
    return part.view(np.uint8).copy()


def vs_partial_from_oracle(batch, threads=1):
    """The 592-byte mi_vs_partial the HIP library would produce for this shard, computed by the oracle (v1 or v2 batch)."""
    if isinstance(batch, VS2Batch):
        _, alg, tot = batch.verify_aggregated(threads)
        return _partial_pack(list(tot), batch.n, int((alg == 0).sum()))
    _, alg, tc, tw = batch.verify_aggregated(threads)
    return _partial_pack([tc, tw], batch.n, int((alg == 0).sum()))


def vs_finalize_with_oracle(vk, partials):
    """Sum the partial accumulators and run the single pairing product: all_ok for the whole batch (v1 contexts: the
    class accumulators are infinity)."""
    parts = np.ascontiguousarray(partials, dtype=np.uint8).reshape(-1, PARTIAL_BYTES).view(np.uint64)
    tc, tw, fails = u64(12), u64(12), 0
    for p in parts:
        for acc, off in ((tc, 0), (tw, 18)):
            pt = p[off:off + 12] if p[off + 12:off + 18].any() else u64(12)
            acc[:] = g1_add_affine(acc, pt)
        fails += int(p[73])
    vkb = np.ascontiguousarray(vk).view(np.uint8).reshape(-1)
    h = np.frombuffer(vkb[96:288].tobytes(), dtype=np.uint64).copy()
    tau_h = np.frombuffer(vkb[288:480].tobytes(), dtype=np.uint64).copy()
    ps = np.stack([tc, tw])
    lib().orc_g1_neg_affine(_p(ps[1]), _p(np.ascontiguousarray(tw)))
    if not tw.any():
        ps[1] = 0
    prep = [C.create_string_buffer(16 + 70 * 192 + 64) for _ in range(2)]
    lib().orc_g2_prepare(prep[0], _p(h))
    lib().orc_g2_prepare(prep[1], _p(tau_h))
    qs = (C.c_void_p * 2)(C.addressof(prep[0]), C.addressof(prep[1]))
    return lib().orc_pairing_product_is_one(_p(ps), qs, C.c_size_t(2)) == 1 and fails == 0


class VSArrays(VSBatch):
    """A VarunaShaped-v1 batch wrapped around existing ABI-layout arrays (e.g. a committed fixture)."""

    def __init__(self, vk, circuits, proofs, randomizers, expected=None, transcript=FS_SHA256):  # noqa: D107
        self.transcript = transcript
        self.vk = np.ascontiguousarray(vk, dtype=np.uint8).reshape(-1)
        self.circuits = np.ascontiguousarray(circuits, dtype=np.uint8).reshape(-1)
        self.proofs = np.ascontiguousarray(proofs, dtype=np.uint8).reshape(-1)
        self.randomizers = np.ascontiguousarray(randomizers, dtype=np.uint8).reshape(-1)
        self.n = self.proofs.size // VS_PROOF_BYTES
        self.n_circuits = self.circuits.size // VS_CIRCUIT_VK_BYTES
        self.expected = expected
        self.seed = None


# ----------------------------------------------------------------------------- Schnorr signatures over Edwards-BLS12 (8f row 4)
SIG_RECORD_BYTES = 288


def ed_generator():
    g = u64(8)
    lib().orc_ed_generator(_p(g))
    return g


def ed_to_ints(p):
    p = np.asarray(p, dtype=np.uint64).reshape(8)
    return fr_to_ints(p[:4])[0], fr_to_ints(p[4:])[0]


def ed_from_ints(pt):
    return np.concatenate([fr_from_ints([pt[0]])[0], fr_from_ints([pt[1]])[0]])


def ed_mul(p, k):
    out = u64(8)
    lib().orc_ed_mul(_p(out), _p(np.ascontiguousarray(p, dtype=np.uint64)), _p(int_to_limbs(k, 4)))
    return out


def ed_hash_to_scalar(inputs_mont, rate):
    """inputs: (n, 4) Montgomery Fr -> the 250-bit scalar"""
    out = u64(4)
    a = np.ascontiguousarray(inputs_mont, dtype=np.uint64).reshape(-1, 4)
    lib().orc_ed_hash_to_scalar(_p(out), _p(a), C.c_int(len(a)), C.c_int(rate))
    return limbs_to_int(out)


class SigBatch:
    """synthetic certificate signatures: record i is signer i % n_signers over message i // n_signers"""

    def __init__(self, seed, n, n_signers=8, corrupt_per_1024=0):
        self.n = n
        self.records = np.zeros(n * SIG_RECORD_BYTES, dtype=np.uint8)
        self.expected = np.zeros(n, dtype=np.uint8)
        lib().orc_schnorr_generate(C.c_uint64(seed), C.c_size_t(n), C.c_size_t(n_signers), C.c_uint32(corrupt_per_1024), _p(self.records),
                                   _p(self.expected))

    def verify_cpu(self, threads=1, records=None):
        r = self.records if records is None else np.ascontiguousarray(records, dtype=np.uint8).reshape(-1)
        v = np.zeros(r.size // SIG_RECORD_BYTES, dtype=np.uint8)
        lib().orc_schnorr_verify_batch(_p(r), C.c_size_t(len(v)), _p(v), C.c_int(threads))
        return v
