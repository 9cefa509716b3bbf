"""CPU tier: the C-ABI library loads, exports every symbol include/mi355x_snark.h declares, its struct
sizes match the header, and (without a GPU) every compute entry point refuses with a status instead of
computing on the CPU: there is no CPU fallback on the product path."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "mi355x_snark.h")).read()
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from snarkos_amd import lib as L
    l = C.CDLL(L.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(l, name), name
    assert sorted(L.EXPORTS) == declared
    assert l.mi_abi_version() == 3


def test_struct_sizes_match_header():
    from snarkos_amd import lib as L
    assert L.G1_AFFINE_U64 * 8 == 96 and L.G1_JAC_U64 * 8 == 144 and L.G2_AFFINE_U64 * 8 == 192
    assert L.KZG_VK_BYTES == 96 + 2 * 192
    assert L.VS_CIRCUIT_VK_BYTES == 12 * 96 + 32
    assert L.VS_PROOF_BYTES == 16 + 16 * 32 + 9 * 96 + 21 * 32 + 2 * 96
    assert L.VS_PARTIAL_BYTES == 4 * 144 + 16 and L.VS2_PROOF_BYTES == L.VS_PROOF_BYTES + 2 * 32 + 9 * 96


def test_error_strings_and_bad_args():
    from snarkos_amd import lib as L
    l = L.lib()
    assert l.mi_error_string(0) == b"ok"
    assert b"device" in l.mi_error_string(-2)
    out = np.zeros(18, dtype=np.uint64)
    assert l.mi_msm_g1(None, None, C.c_size_t(4), L.ptr(out)) == -1       # MI_ERR_BAD_ARGS
    assert l.mi_ntt_fr(None, C.c_uint32(4), 0, 0) == -1
    buf = np.zeros(4, dtype=np.uint64)
    assert l.mi_ntt_fr(L.ptr(buf), C.c_uint32(31), 0, 0) == -1            # domain too large
    assert l.mi_msm_g1(L.ptr(buf), L.ptr(buf), C.c_size_t(1 << 28), L.ptr(out)) == -1  # more entries than 32-bit indices hold


def test_no_cpu_fallback_without_a_gpu():
    """On a box without a GPU the product must fail loudly (MI_ERR_NO_DEVICE), never compute on the host."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present")
    from snarkos_amd import algorithms as A, lib as L
    bases = np.zeros((4, 12), dtype=np.uint64)
    scalars = np.zeros((4, 4), dtype=np.uint64)
    try:
        A.VariableBase.msm(bases, scalars)
        raise AssertionError("msm computed without a device")
    except L.MiError as e:
        assert e.status in (-2, -4)


def test_product_never_imports_the_oracle():
    """static check: nothing under snarkos_amd/ or include/ references oracle/"""
    for base in ("snarkos_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            if "_obj" in dp or "__pycache__" in dp:
                continue
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h")):
                    txt = open(os.path.join(dp, f)).read()
                    assert "orc_" not in txt and "liboracle" not in txt and "pyref" not in txt, os.path.join(dp, f)
