"""ctypes loader for libmi355x_snark.so (the C-ABI of include/mi355x_snark.h).

The product path has NO CPU fallback: if the HIP library is missing or a call returns a non-zero
status, a MiError is raised.  (A Rust caller would fall back to snarkVM's own CPU path on a non-zero
status; this Python mirror exists for tests and benchmarks, where a silent fallback would hide bugs.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI_SNARK_LIB: another build of the same library (A/B measurements of one translation unit: tools/ab_lib.sh)
LIB_PATH = os.environ.get("MI_SNARK_LIB") or os.path.join(_HERE, "libmi355x_snark.so")

MI_OK = 0
MI_ERR_NO_DEVICE = -2
MI_ERR_KERNEL = -4
MI_ERR_AGAIN = -6
MI_ERR_DONE = -7
# int32_t (*mi_vs_exchange_fn)(void *user, const mi_vs_partial *mine, size_t count, mi_vs_partial *all, size_t max_partials)
VS_EXCHANGE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_uint8), C.c_size_t)
G1_AFFINE_U64 = 12
G1_JAC_U64 = 18
G2_AFFINE_U64 = 24
FQ12_U64 = 72
KZG_VK_BYTES = 480
VS_PROOF_BYTES = 2256
VS_CIRCUIT_VK_BYTES = 1184
VS_PARTIAL_BYTES = 592
VS_MAX_OPEN_ACCUMULATIONS = 14  # compile-time cap; a device has MI_SNARK_VS_LANES (default 12) lanes
VS_DEFAULT_LANES = 12
VS_MAX_TICKETS = 16  # compile-time cap; a device has MI_SNARK_VS_SLOTS (default 8) product slots
VS_FS_POSEIDON = 1  # mi_vs_ctx_create_ex flag: Poseidon Fiat-Shamir transcript
VS2_PROOF_BYTES = 3184  # mi_vs_proof2
VS3_PROOF_BYTES = 5504  # mi_vs_proof3 (shapes with max_instances > 1)
VS4_PROOF_BYTES = 6320  # mi_vs_proof4 (shapes with max_circuits > 1)
VS_WIRE = 2  # mi_vs_ctx_create_* flag: records arrive in the wire form (compressed points, canonical Fr)
VS2_KEY_EXT_BYTES = 480
VS2_SHAPE_BYTES = 56
SIG_RECORD_BYTES = 288  # mi_sig_record

# every symbol include/mi355x_snark.h declares
EXPORTS = [
    "mi_init", "mi_device_count", "mi_device_id", "mi_shutdown", "mi_error_string", "mi_abi_version", "mi_runtime_hints", "mi_stats_read",
    "mi_vs_strides", "mi_vs_wire_encode",
    "mi_msm_g1", "mi_msm_g1_device", "mi_g1_sum", "mi_bases_register", "mi_bases_release", "mi_msm_g1_h",
    "mi_ntt_fr", "mi_ntt_fr_device", "mi_kzg_commit_h", "mi_kzg_commit_h_device", "mi_kzg_open_h", "mi_poly_mul_fr",
    "mi_kzg_commit_hiding_h", "mi_kzg_open_hiding_h", "mi_kzg_commit_lagrange_h",
    "mi_ntt_divide_by_vanishing_on_coset", "mi_ntt_divide_by_vanishing_on_coset_device",
    "mi_pairing_product_is_one", "mi_pairing", "mi_kzg_batch_check", "mi_kzg_batch_check_hiding",
    "mi_poseidon_fq_hash_many", "mi_poseidon_fq_hash_many_device", "mi_schnorr_verify_batch", "mi_schnorr_verify_batch_device",
    "mi_vs_ctx_create", "mi_vs_ctx_create_ex", "mi_vs_ctx_create_v2", "mi_vs_ctx_destroy", "mi_vs_verify_batch", "mi_vs_verify_batch_device", "mi_vs_identify_device",
    "mi_vs_accumulate_device", "mi_vs_accumulate_begin", "mi_vs_accumulate_end", "mi_vs_accumulate_begin_pair", "mi_vs_accumulate_end_pair", "mi_vs_accumulate_begin_multi", "mi_vs_accumulate_end_multi", "mi_vs_accumulate_begin_merged", "mi_vs_batches_per_pass", "mi_vs_batches_per_merged_pass", "mi_g1_decompress", "mi_g1_decompress_device", "mi_g1_compress", "mi_vs_finalize", "mi_vs_finalize_async", "mi_vs_finalize_wait", "mi_vs_finalize_async_multi", "mi_vs_finalize_wait_multi", "mi_vs_wait_product_slot",
    "mi_vs_verify_block", "mi_vs_verify_block_device",
    "mi_vs_stream_open", "mi_vs_stream_submit", "mi_vs_stream_finish", "mi_vs_stream_next", "mi_vs_stream_next_verdicts", "mi_vs_stream_close",
    "mi_comm_unique_id", "mi_comm_init_rank", "mi_comm_init_local", "mi_comm_world", "mi_comm_transport", "mi_comm_destroy", "mi_comm_all_gather",
    "mi_vs_exchange_rccl",
    "mi_dbg_fq_mul", "mi_dbg_fq_mul28", "mi_dbg_g1_add28", "mi_dbg_fr_mul", "mi_dbg_g1_add", "mi_dbg_fq_mul_throughput", "mi_dbg_gen_points_device", "mi_dbg_msm_plan",
    "mi_prof_enable", "mi_prof_read",
]


class MiError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        try:
            msg = lib().mi_error_string(status).decode()
        except Exception:  # pragma: no cover
            msg = "?"
        super().__init__("%s failed: status %d (%s)" % (where, status, msg))


_LIB = None


def lib():
    """Load the HIP library; fail loudly if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libmi355x_snark.so is not built (run `python -m snarkos_amd.build`); "
                "there is no CPU fallback on the product path")
        # The PyTorch wheel bundles its own libamdhip64.so.7 / libhsa-runtime64; two HIP runtimes in one
        # process cannot both own the GPU ("No HIP GPUs are available" for whichever comes second).
        # Importing torch first makes the dynamic loader resolve our NEEDED libamdhip64.so.7 to the copy
        # torch already mapped, so device pointers and streams are shared with torch tensors.
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is plumbing, not a requirement of the C ABI
            pass
        l = C.CDLL(LIB_PATH)
        for name in EXPORTS:
            getattr(l, name).restype = C.c_int32
        l.mi_error_string.restype = C.c_char_p
        l.mi_vs_stream_open.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_int32, VS_EXCHANGE_FN, C.c_void_p, C.c_int32, C.POINTER(C.c_uint64)]
        l.mi_vs_stream_submit.argtypes = [C.c_uint64, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        l.mi_vs_stream_next.argtypes = [C.c_uint64, C.c_int32, C.POINTER(C.c_uint8)]
        l.mi_vs_stream_next_verdicts.argtypes = [C.c_uint64, C.c_int32, C.POINTER(C.c_uint8), C.c_void_p, C.c_size_t,
                                                 C.POINTER(C.c_size_t)]
        l.mi_vs_stream_finish.argtypes = [C.c_uint64]
        l.mi_vs_stream_close.argtypes = [C.c_uint64]
        l.mi_error_string.argtypes = [C.c_int32]
        l.mi_init.argtypes = [C.POINTER(C.c_int32), C.c_int32]
        l.mi_device_id.argtypes = [C.c_int32]
        l.mi_comm_unique_id.argtypes = [C.c_void_p]
        l.mi_comm_init_rank.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_uint64)]
        l.mi_comm_init_local.argtypes = [C.POINTER(C.c_uint64)]
        l.mi_comm_world.argtypes = [C.c_uint64]
        l.mi_comm_transport.argtypes = [C.c_uint64]
        l.mi_comm_destroy.argtypes = [C.c_uint64]
        l.mi_comm_all_gather.argtypes = [C.c_uint64, C.c_void_p, C.c_size_t, C.c_void_p]
        l.mi_vs_verify_block.argtypes = [C.c_uint64, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        l.mi_vs_verify_block_device.argtypes = [C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                                C.c_int32, C.c_void_p, C.c_void_p]
        _LIB = l
    return _LIB


def check(status, where):
    if status != MI_OK:
        raise MiError(status, where)


def ptr(a):
    """numpy array -> void* (must be C-contiguous)"""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
    return a.ctypes.data_as(C.c_void_p)


def dptr(t):
    """torch CUDA tensor (or int address) -> void*"""
    if isinstance(t, int):
        return C.c_void_p(t)
    assert t.is_cuda and t.is_contiguous()
    return C.c_void_p(t.data_ptr())


def stream_ptr(stream=None):
    """torch.cuda.Stream (or None = torch's current stream) -> hipStream_t as void*"""
    import torch

    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def init(devices=None):
    """mi_init(device_ids, n): `devices` = None (the current HIP device), one ordinal, or a list whose first entry
    becomes the primary device.  Idempotent; later calls may add devices."""
    if devices is None or (isinstance(devices, int) and devices < 0):
        check(lib().mi_init(None, 0), "mi_init")
        return
    ids = [int(devices)] if isinstance(devices, int) else [int(d) for d in devices]
    arr = (C.c_int32 * len(ids))(*ids)
    check(lib().mi_init(arr, len(ids)), "mi_init")


def device_count():
    return int(lib().mi_device_count())


COMM_ID_BYTES = 128


def comm_unique_id():
    """128 opaque bytes to hand to every rank (mi_comm_unique_id); raises MiError(MI_ERR_NO_DEVICE) without RCCL."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    check(lib().mi_comm_unique_id(buf), "mi_comm_unique_id")
    return bytes(buf)


def comm_init_rank(uid, rank, world):
    """Collective: every rank calls it with rank 0's id.  Returns the communicator handle."""
    assert len(uid) == COMM_ID_BYTES
    h = C.c_uint64(0)
    buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(uid)
    check(lib().mi_comm_init_rank(buf, C.c_int32(rank), C.c_int32(world), C.byref(h)), "mi_comm_init_rank")
    return h.value


def comm_transport(comm):
    """"rccl" or "host" (shared memory between the node's ranks): decided by all ranks together at comm_init_rank"""
    t = lib().mi_comm_transport(C.c_uint64(comm))
    if t < 0:
        raise MiError(t, "mi_comm_transport")
    return {1: "rccl", 2: "host"}[t]


def comm_destroy(comm):
    check(lib().mi_comm_destroy(C.c_uint64(comm)), "mi_comm_destroy")


def shutdown():
    check(lib().mi_shutdown(), "mi_shutdown")


STATS_FIELDS = ("batches_verified", "proofs_verified", "proofs_rejected", "verify_batch_nanos", "msm_calls", "msm_points", "ntt_calls",
                "ntt_elements", "pairing_products", "pairing_pairs", "points_decompressed", "fallback_batches", "exchange_calls", "exchange_nanos")


def stats_read():
    """mi_stats_read -> dict of the monotonic counters"""
    buf = np.zeros(16, dtype=np.uint64)
    check(lib().mi_stats_read(ptr(buf)), "mi_stats_read")
    return {k: int(buf[i]) for i, k in enumerate(STATS_FIELDS)}


def runtime_hints():
    return int(lib().mi_runtime_hints())


def prof_enable(level=1):
    """0 / False: off; 1 / True: headline kernels only; 2: every section"""
    check(lib().mi_prof_enable(C.c_int32(int(level))), "mi_prof_enable")


def prof_read(kernel):
    """-> (total_ms, launches, units) of the HIP-event records of `kernel`, then clears them"""
    ms, n, u = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
    check(lib().mi_prof_read(kernel.encode(), C.byref(ms), C.byref(n), C.byref(u)), "mi_prof_read")
    return ms.value, n.value, u.value
