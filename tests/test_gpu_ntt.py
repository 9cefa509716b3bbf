"""-m gpu: EvaluationDomain::{fft,ifft,coset_fft,coset_ifft} parity against the CPU oracle."""
import numpy as np
import pytest

from helpers import rng_fr_mont

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 4, 5, 7, 8, 9, 12, 13, 16])
def test_ntt_matches_oracle(orc, mi, log_n):
    n = 1 << log_n
    a = rng_fr_mont(300 + log_n, n)
    d = mi.EvaluationDomain(n)
    for inverse, coset, fn in ((0, 0, d.fft), (1, 0, d.ifft), (0, 1, d.coset_fft), (1, 1, d.coset_ifft)):
        got = fn(a)
        want = orc.ntt(a, inverse=bool(inverse), coset=bool(coset), threads=8)
        assert (got == want).all(), (log_n, inverse, coset)


def test_ntt_zero_padding(orc, mi):
    a = rng_fr_mont(5, 100)
    d = mi.EvaluationDomain(100)
    assert d.size == 128
    padded = np.zeros((128, 4), dtype=np.uint64); padded[:100] = a
    assert (d.fft(a) == orc.ntt(padded)).all()


def test_ntt_roundtrip_large(orc, mi):
    """size-independent property at 2^20: ifft(fft(a)) == a bit-exact, plus Horner spot checks"""
    import pyref
    log_n = 20
    n = 1 << log_n
    a = rng_fr_mont(99, n)
    d = mi.EvaluationDomain(n)
    f = d.fft(a)
    assert (d.ifft(f) == a).all()
    assert (d.coset_ifft(d.coset_fft(a)) == a).all()
    # f[j] == a(w^j) for a sparse polynomial (only the first 64 coefficients non-zero)
    b = np.zeros_like(a); b[:64] = a[:64]
    fb = d.fft(b)
    w = pyref.omega(log_n)
    coeffs = orc.fr_to_ints(b[:64])
    for j in (0, 1, 12345, n // 2, n - 1):
        x = pow(w, j, pyref.R)
        want = sum(c * pow(x, i, pyref.R) for i, c in enumerate(coeffs)) % pyref.R
        assert orc.fr_to_ints(fb[j:j + 1])[0] == want


def test_ntt_full_size_2_22_vs_oracle(orc, mi):
    """BASELINE configs[2]: domain 2^22, forward transform bit-exact against the CPU oracle, then the round trip"""
    log_n = 22
    n = 1 << log_n
    a = rng_fr_mont(2022, n)
    d = mi.EvaluationDomain(n)
    f = d.fft(a)
    assert (f == orc.ntt(a, threads=16)).all()
    assert (d.ifft(f) == a).all()


@pytest.mark.parametrize("log_n", [11, 12, 13, 14, 15, 17, 18, 19, 20, 21])
def test_ntt_lds_passes_all_splits(orc, mi, log_n):
    """every pass-width combination of the LDS-tiled kernel (4..9 bits per pass), all four transform flavours"""
    n = 1 << log_n
    a = rng_fr_mont(700 + log_n, n)
    d = mi.EvaluationDomain(n)
    for inverse, coset, fn in ((0, 0, d.fft), (1, 0, d.ifft), (0, 1, d.coset_fft), (1, 1, d.coset_ifft)):
        if log_n > 17 and coset:
            continue  # the oracle's coset scaling is serial; the coset kernels do not depend on the pass split
        assert (fn(a) == orc.ntt(a, inverse=bool(inverse), coset=bool(coset), threads=16)).all(), (log_n, inverse, coset)
