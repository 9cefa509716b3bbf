"""-m gpu: device field and group arithmetic, bit-exact against the CPU oracle (through the C ABI)."""
import ctypes as C

import numpy as np
import pytest

from helpers import rng_points, rng_scalars

pytestmark = pytest.mark.gpu


def test_fq_mul_bit_exact(orc, mi):
    from snarkos_amd import lib as L
    n = 4096
    g = np.random.default_rng(1)
    a = np.zeros((n, 6), dtype=np.uint64)
    b = np.zeros((n, 6), dtype=np.uint64)
    for arr, s in ((a, 11), (b, 12)):
        pts = rng_points(s, n // 2)  # coordinates are uniform-looking field elements < q
        arr[:] = pts.reshape(-1, 6)
    a[0] = 0
    b[1] = 0
    a[2] = np.array(orc.int_to_limbs(orc_q() - 1, 6))
    b[2] = a[2]
    out = np.zeros_like(a)
    L.check(L.lib().mi_dbg_fq_mul(L.ptr(a), L.ptr(b), L.ptr(out), C.c_size_t(n)), "mi_dbg_fq_mul")
    want = np.zeros_like(a)
    for i in range(n):
        orc.lib().orc_fq_mul(orc._p(want[i]), orc._p(a[i]), orc._p(b[i]))
    assert (out == want).all()


def orc_q():
    import pyref
    return pyref.Q


def test_fr_mul_bit_exact(orc, mi):
    from snarkos_amd import lib as L
    n = 4096
    a, b = rng_scalars(21, n), rng_scalars(22, n)
    a[0] = 0
    out = np.zeros_like(a)
    L.check(L.lib().mi_dbg_fr_mul(L.ptr(a), L.ptr(b), L.ptr(out), C.c_size_t(n)), "mi_dbg_fr_mul")
    want = np.zeros_like(a)
    for i in range(n):
        orc.lib().orc_fr_mul(orc._p(want[i]), orc._p(a[i]), orc._p(b[i]))
    assert (out == want).all()


def test_g1_add_all_cases(orc, mi):
    from snarkos_amd import lib as L
    n = 256
    a, b = rng_points(31, n), rng_points(32, n)
    b[0] = a[0]                      # doubling
    b[1] = a[1]; b[1, 6:] = neg_y(orc, a[1, 6:])  # P + (-P) = O
    a[2] = 0                         # O + Q
    b[3] = 0                         # P + O
    a[4] = 0; b[4] = 0               # O + O
    out = np.zeros_like(a)
    L.check(L.lib().mi_dbg_g1_add(L.ptr(a), L.ptr(b), L.ptr(out), C.c_size_t(n)), "mi_dbg_g1_add")
    for i in range(n):
        j = np.zeros(18, dtype=np.uint64)
        ja = np.zeros(18, dtype=np.uint64)
        orc.lib().orc_g1_from_affine(orc._p(ja), orc._p(np.ascontiguousarray(a[i])))
        orc.lib().orc_g1_add_mixed(orc._p(j), orc._p(ja), orc._p(np.ascontiguousarray(b[i])))
        assert (orc.jac_to_affine(j) == out[i]).all(), i


def neg_y(orc, y):
    import pyref
    v = orc.fq_to_int(y)
    return orc.fq_from_int((-v) % pyref.Q)


def test_entry_points_are_thread_safe(orc, mi):
    """callers are rayon / tokio threads: concurrent calls from several host threads must all be exact"""
    import threading
    from helpers import jac_affine, rng_fr_mont
    bases, scalars = rng_points(71, 600), rng_scalars(72, 600)
    want_msm = orc.msm_pippenger(bases, scalars, threads=4)
    a = rng_fr_mont(73, 1 << 10)
    want_ntt = orc.ntt(a, threads=4)
    errors = []

    def worker(kind):
        try:
            for _ in range(4):
                if kind == 0:
                    assert (jac_affine(mi.VariableBase.msm(bases, scalars)) == want_msm).all()
                else:
                    assert (mi.EvaluationDomain(1 << 10).fft(a) == want_ntt).all()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(i % 2,)) for i in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_device_entry_points_on_two_streams_from_two_threads(orc, mi):
    """the asynchronous `_device` entry points leave their kernels running on the CALLER's stream: two host threads on
    two streams must not share scratch (round 1's name-keyed "ntt.tmp" did), whatever the order of sizes -- a larger
    transform arriving while a smaller one is in flight regrows nothing that is in use.  Bit-exact against the oracle;
    an MSM on a third stream runs beside them."""
    import threading
    import torch
    from helpers import jac_affine, rng_fr_mont
    sizes = [12, 16, 10, 17, 9, 14]
    data = {lg: rng_fr_mont(900 + lg, 1 << lg) for lg in set(sizes)}
    want = {lg: orc.ntt(data[lg], threads=4) for lg in data}
    bases, scalars = rng_points(91, 3000), rng_scalars(92, 3000)
    want_msm = orc.msm_pippenger(bases, scalars, threads=4)
    d_bases, d_scalars = torch.from_numpy(bases).cuda(), torch.from_numpy(scalars.view(np.int64)).cuda()
    errors = []

    def ntt_worker(order):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                bufs = []
                for rep in range(3):
                    for lg in order:
                        t = torch.from_numpy(data[lg].view(np.int64)).cuda()
                        mi.EvaluationDomain(1 << lg).fft(t, stream=st)  # returns with the passes still in flight
                        bufs.append((lg, t))
                st.synchronize()
                for lg, t in bufs:
                    assert (t.cpu().numpy().view(np.uint64) == want[lg]).all(), lg
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def msm_worker():
        try:
            st = torch.cuda.Stream()
            for _ in range(4):
                assert (jac_affine(mi.VariableBase.msm(d_bases, d_scalars, stream=st)) == want_msm).all()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=ntt_worker, args=(sizes,)), threading.Thread(target=ntt_worker, args=(sizes[::-1],)),
               threading.Thread(target=msm_worker)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_reduced_radix_field_arithmetic_bit_exact(orc, mi):
    """field28.hpp (14 signed 28-bit limbs) through its conversions: products, squarings and lazy additions against the
    oracle's Montgomery product; edge operands: 0, 1, q - 1, the Montgomery one, all-ones limb patterns below q"""
    import pyref
    from snarkos_amd import lib as L
    n = 4096
    a = rng_points(41, n // 2).reshape(-1, 6).copy()
    b = rng_points(42, n // 2).reshape(-1, 6).copy()
    q = pyref.Q
    edge = [0, 1, q - 1, q - 2, (1 << 384) % q, (1 << 376) - 1, int("f" * 7 + "0" * 7, 16) << 300, (1 << 28) - 1, 1 << 28, (1 << 56) + 5]
    k = 0
    for x in edge:
        for y in edge:
            a[k] = orc.int_to_limbs(x % q, 6); b[k] = orc.int_to_limbs(y % q, 6)
            a[k + 1] = a[k]; b[k + 1] = b[k]       # the same pair on an odd lane
            k += 2
    out = np.zeros_like(a)
    L.check(L.lib().mi_dbg_fq_mul28(L.ptr(a), L.ptr(b), L.ptr(out), C.c_size_t(n)), "mi_dbg_fq_mul28")
    rinv = pow(1 << 384, -1, q)
    for i in range(n):
        x, y = orc.limbs_to_int(a[i]), orc.limbs_to_int(b[i])
        want = x * y * rinv % q
        if i & 1:
            want = 4 * want % q
        assert orc.limbs_to_int(out[i]) == want, i


def test_reduced_radix_group_law_all_cases(orc, mi):
    """add_mixed28 / xyzz_add28 / their degenerate branches: out = a + b computed as ((a + b) + b) - b"""
    from snarkos_amd import lib as L
    n = 256
    a, b = rng_points(51, n), rng_points(52, n)
    b[0] = a[0]                                        # doubling inside the mixed addition
    b[1] = a[1]; b[1, 6:] = neg_y(orc, a[1, 6:])       # a + b = O, then O + b, then b - b = O
    a[2] = 0
    b[3] = 0
    a[4] = 0; b[4] = 0
    out = np.zeros_like(a)
    L.check(L.lib().mi_dbg_g1_add28(L.ptr(a), L.ptr(b), L.ptr(out), C.c_size_t(n)), "mi_dbg_g1_add28")
    for i in range(n):
        want = orc.g1_add_affine(a[i], b[i])
        assert (out[i] == want).all(), i
