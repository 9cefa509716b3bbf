"""-m gpu: VariableBase::msm parity (bit-exact canonical affine) against the CPU oracle."""
import numpy as np
import pytest

from helpers import jac_affine, rng_points, rng_scalars

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 2, 14, 15, 100, 1000, 5000, 1 << 14])
def test_msm_matches_oracle(orc, mi, n):
    bases, scalars = rng_points(100 + n, n), rng_scalars(200 + n, n)
    got = jac_affine(mi.VariableBase.msm(bases, scalars))
    want = orc.msm_pippenger(bases, scalars, threads=8)
    assert (got == want).all()


def test_msm_edge_cases(orc, mi):
    import pyref
    n = 300
    bases, scalars = rng_points(7, n), rng_scalars(8, n)
    scalars[0] = 0                                        # zero scalar
    scalars[1] = orc.int_to_limbs(pyref.R - 1, 4)         # r - 1
    scalars[2] = orc.int_to_limbs(1, 4)
    bases[3] = 0                                          # point at infinity
    bases[5] = bases[4]; scalars[5] = scalars[4]          # repeated pair (same bucket: doubling)
    bases[7] = bases[6]; bases[7, 6:] = orc.fq_from_int((-orc.fq_to_int(bases[6, 6:])) % pyref.Q)
    scalars[7] = scalars[6]                               # P and -P with equal scalars cancel
    got = jac_affine(mi.VariableBase.msm(bases, scalars))
    assert (got == orc.msm_pippenger(bases, scalars, threads=8)).all()
    # all-zero scalars -> infinity; empty input -> infinity
    z = np.zeros_like(scalars)
    assert not jac_affine(mi.VariableBase.msm(bases, z)).any()
    assert not jac_affine(mi.VariableBase.msm(bases[:0], scalars[:0])).any()
    # min(len) truncation
    got = jac_affine(mi.VariableBase.msm(bases[:50], scalars))
    assert (got == orc.msm_pippenger(bases[:50], scalars[:50], threads=8)).all()


def test_msm_all_scalars_equal(orc, mi):
    """every point lands in the same bucket of each window"""
    n = 2000
    bases = rng_points(9, n)
    scalars = np.tile(rng_scalars(10, 1), (n, 1))
    got = jac_affine(mi.VariableBase.msm(bases, scalars))
    assert (got == orc.msm_pippenger(bases, scalars, threads=8)).all()


def test_msm_linearity_large(orc, mi):
    """size-independent property at 2^18: msm(P, a) + msm(P, b) == msm(P, a + b mod r)"""
    import pyref
    n = 1 << 18
    small = rng_points(77, 1 << 10)
    bases = np.tile(small, (n >> 10, 1))
    a, b = rng_scalars(78, n), rng_scalars(79, n)
    s = np.zeros_like(a)
    for i in range(0, n, 4096):  # only a sample needs exact modular sums; the rest reuse it
        pass
    ai = [orc.limbs_to_int(x) for x in a[:1024]]
    bi = [orc.limbs_to_int(x) for x in b[:1024]]
    a = np.tile(a[:1024], (n >> 10, 1)); b = np.tile(b[:1024], (n >> 10, 1))
    s = np.tile(np.stack([orc.int_to_limbs((x + y) % pyref.R, 4) for x, y in zip(ai, bi)]), (n >> 10, 1))
    ra = jac_affine(mi.VariableBase.msm(bases, a)); rb = jac_affine(mi.VariableBase.msm(bases, b))
    rs = jac_affine(mi.VariableBase.msm(bases, s))
    ja = np.zeros(18, dtype=np.uint64); j = np.zeros(18, dtype=np.uint64)
    orc.lib().orc_g1_from_affine(orc._p(ja), orc._p(ra))
    orc.lib().orc_g1_add_mixed(orc._p(j), orc._p(ja), orc._p(rb))
    assert (orc.jac_to_affine(j) == rs).all()


def test_msm_resident_bases(orc, mi):
    n = 3000
    bases, scalars = rng_points(41, n), rng_scalars(42, n)
    rb = mi.ResidentBases(bases)
    try:
        assert (jac_affine(rb.msm(scalars)) == orc.msm_pippenger(bases, scalars, threads=8)).all()
        assert (jac_affine(rb.msm(scalars[:1000], offset=500)) == orc.msm_pippenger(bases[500:1500], scalars[:1000], threads=8)).all()
    finally:
        rb.release()


def test_msm_full_size_2_20_bit_exact(orc, mi):
    """BASELINE configs[1]: 2^20-point MSM on the GPU, bit-exact against the CPU oracle's Pippenger"""
    import ctypes as C
    import torch
    from snarkos_amd import lib as L
    n = 1 << 20
    pts = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
    L.check(L.lib().mi_dbg_gen_points_device(C.c_uint64(0xA1E00002), C.c_size_t(n), L.dptr(pts), L.stream_ptr()), "gen_points")
    scalars = rng_scalars(0xA1E00002, n)
    d_sc = torch.from_numpy(scalars.view(np.int64)).cuda()
    got = jac_affine(mi.VariableBase.msm(pts, d_sc))
    bases = pts.cpu().numpy().view(np.uint64).reshape(n, 12)
    # generated points are on the curve (spot check) and the oracle agrees on the sum
    for i in (0, 1, n // 2, n - 1):
        assert orc.lib().orc_g1_is_on_curve(orc._p(np.ascontiguousarray(bases[i])))
    want = orc.msm_pippenger(bases, scalars, threads=16)
    assert (got == want).all()


def test_msm_skewed_scalars_large(orc, mi):
    """adversarial load: 2^16 points, only 3 distinct scalars (three buckets per window hold everything)"""
    n = 1 << 16
    bases = rng_points(55, 1 << 10)
    bases = np.tile(bases, (n >> 10, 1))
    three = rng_scalars(56, 3)
    scalars = three[np.arange(n) % 3]
    got = jac_affine(mi.VariableBase.msm(bases, np.ascontiguousarray(scalars)))
    assert (got == orc.msm_pippenger(bases, np.ascontiguousarray(scalars), threads=16)).all()


def test_msm_points_split_across_ranks(orc, mi):
    """one MSM with its points cut into shards (here: three 'ranks' on one GPU, uneven, one of them
    empty-handed), partial sums combined by mi_g1_sum: bit-exact with the unsharded MSM and with the oracle."""
    from snarkos_amd import sharding
    n = 5000
    bases, scalars = rng_points(31, n), rng_scalars(32, n)
    parts = []
    for r in range(3):
        lo, hi = sharding.shard_bounds(n, 3, r)
        parts.append(mi.VariableBase.msm(bases[lo:hi], scalars[lo:hi]))
    parts.append(mi.VariableBase.msm(bases[:0], scalars[:0]))           # infinity partial
    neg = bases[:1].copy(); neg[0, 6:] = orc.fq_from_int((-orc.fq_to_int(bases[0, 6:])) % __import__("pyref").Q)
    parts.append(mi.VariableBase.msm(bases[:1], scalars[:1]))           # P and -P: cancel
    parts.append(mi.VariableBase.msm(neg, scalars[:1]))
    got = jac_affine(mi.VariableBase.sum_partials(np.stack(parts)))
    assert (got == jac_affine(mi.VariableBase.msm(bases, scalars))).all()
    assert (got == orc.msm_pippenger(bases, scalars, threads=8)).all()
    # doubling inside the sum (the same partial twice) and the empty sum
    twice = jac_affine(mi.VariableBase.sum_partials(np.stack([parts[0], parts[0]])))
    assert (twice == orc.g1_add_affine(jac_affine(parts[0]), jac_affine(parts[0]))).all()
    assert not jac_affine(mi.VariableBase.sum_partials(np.zeros((0, 18), dtype=np.uint64))).any()
    # the protocol object, world = 1
    whole = sharding.msm_sharded(bases, scalars, mi.VariableBase.msm, mi.VariableBase.sum_partials)
    assert (jac_affine(whole) == got).all()


@pytest.mark.parametrize("n", [(1 << 16) - 1, 1 << 16, (1 << 16) + 1])
def test_msm_around_the_small_path_limit(orc, mi, n):
    """the last sizes of the latency-bound path (fused planning kernel, bit sums) and the first of the general one,
    with a block of zero scalars, repeated points and a run of equal scalars (one bucket shared by a whole wave:
    the aggregated histogram atomics)"""
    bases, scalars = rng_points(7000 + n % 7, 4096), rng_scalars(7100 + n % 7, n)
    bases = np.ascontiguousarray(np.tile(bases, (n // 4096 + 1, 1))[:n])   # repeated points
    scalars[100:400] = 0
    scalars[1000:1200] = scalars[1000]                                      # 200 equal scalars in a row
    got = jac_affine(mi.VariableBase.msm(bases, scalars))
    assert (got == orc.msm_pippenger(bases, scalars, threads=8)).all()


def test_msm_2_21_exact_by_known_discrete_logs(orc, mi):
    """beyond what a CPU Pippenger checks in seconds: the generated points are k_i * G with k_i = splitmix64(seed + i) | 1,
    so the MSM must equal (sum_i s_i k_i mod r) * G -- a host big-integer dot product and one oracle scalar
    multiplication (the same check passes at 2^22 and 2^24: tools/msm_big_check.py)"""
    import ctypes as C
    import torch
    import pyref
    from snarkos_amd import lib as L
    n, seed = 1 << 21, 0xB17
    pts = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
    L.check(L.lib().mi_dbg_gen_points_device(C.c_uint64(seed), C.c_size_t(n), L.dptr(pts), L.stream_ptr()), "gen_points")
    scalars = rng_scalars(0xB18, n)
    got = jac_affine(mi.VariableBase.msm(pts, torch.from_numpy(scalars.view(np.int64)).cuda()))
    M = 0xFFFFFFFFFFFFFFFF

    def splitmix(x):
        x = (x + 0x9E3779B97F4A7C15) & M
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M
        return x ^ (x >> 31)
    lim = scalars.astype(object)
    t = 0
    for i in range(n):
        s = lim[i, 0] | (lim[i, 1] << 64) | (lim[i, 2] << 128) | (lim[i, 3] << 192)
        t += s * (splitmix(seed + i) | 1)
    assert (got == orc.g1_mul(orc.g1_generator(), t % pyref.R)).all()


def _uniform_scalars(seed, n):
    """canonical scalars uniform in [0, r), (n, 4) uint64: rejection sampling on 253 bits"""
    import pyref
    g = np.random.default_rng(seed)
    out = np.zeros((n, 4), dtype=np.uint64)
    r_limbs = [(pyref.R >> (64 * k)) & ((1 << 64) - 1) for k in range(4)]
    todo = np.arange(n)
    while len(todo):
        a = g.integers(0, 1 << 63, size=(len(todo), 4), dtype=np.uint64) * np.uint64(2) + g.integers(0, 2, size=(len(todo), 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 61) - 1)  # 253 bits
        lt = np.zeros(len(todo), dtype=bool)
        eq = np.ones(len(todo), dtype=bool)
        for k in (3, 2, 1, 0):
            lt |= eq & (a[:, k] < np.uint64(r_limbs[k]))
            eq &= a[:, k] == np.uint64(r_limbs[k])
        out[todo[lt]] = a[lt]
        todo = todo[~lt]
    return out


@pytest.mark.parametrize("log_n", [15, 17, 18, 20])
def test_msm_full_range_scalars_large_path(orc, mi, log_n):
    """scalars uniform in [0, r) -- a 253-bit top digit in the last window -- with r - 1, r - 2, 2^252 and the window
    boundaries planted, on both MSM paths (2^15: small / per-bit; 2^17..2^20: tiled-histogram large path, one and two
    bits of window-width bias).  Bit-exact against the oracle's Pippenger."""
    import pyref
    n = 1 << log_n
    small = rng_points(300 + log_n, 1 << 12)
    bases = np.tile(small, (n >> 12, 1))
    scalars = _uniform_scalars(400 + log_n, n)
    assert (scalars[:, 3] >> np.uint64(60)).any()  # the 253rd bit is in use
    plant = [pyref.R - 1, pyref.R - 2, 1 << 252, (1 << 252) - 1, pyref.R >> 1, (1 << 253) - (1 << 200)]
    for c in (13, 14, 15, 16, 17):  # every window at its extreme digits for the widths the planner may pick
        plant.append(sum(((1 << (c - 1)) << (c * w)) for w in range(253 // c)) % pyref.R)
        plant.append(sum((((1 << c) - 1) << (c * w)) for w in range(253 // c)) % pyref.R)
    for k, x in enumerate(plant):
        scalars[(k * 7919) % n] = orc.int_to_limbs(x % pyref.R, 4)
    got = jac_affine(mi.VariableBase.msm(bases, scalars))
    want = orc.msm_pippenger(bases, scalars, threads=16)
    assert (got == want).all()


@pytest.mark.parametrize("n", [64, 3000, 1 << 17])
def test_msm_degenerate_additions_in_the_accumulation(orc, mi, n):
    """the bucket accumulation's own degenerate branches (reduced-radix arithmetic, field28.hpp add_mixed28): many
    copies of ONE point (every bucket addition after the first two is P + kP, the second is a doubling), runs of P, -P
    with equal scalars (sums that pass through infinity inside a chunk), and both mixed into random data; small and
    large path"""
    import pyref
    base = rng_points(31, 4)
    negb = base.copy()
    for i in range(4):
        negb[i, 6:] = orc.fq_from_int((-orc.fq_to_int(base[i, 6:])) % pyref.Q)
    sc = rng_scalars(32, 8)
    bases, scalars = rng_points(33, n), rng_scalars(34, n)
    bases[: n // 4] = base[0]                      # one point, random scalars
    scalars[n // 4: n // 2] = sc[0]                # random points, one scalar
    q = n // 2
    for j in range(q, q + (n // 8)):               # P, -P, P, -P ... with one scalar: same buckets, alternating
        bases[j] = base[1] if (j & 1) == 0 else negb[1]
        scalars[j] = sc[1]
    for j in range(q + n // 8, q + n // 4):        # P, P, -P, -P ...
        bases[j] = base[2] if ((j >> 1) & 1) == 0 else negb[2]
        scalars[j] = sc[2]
    got = jac_affine(mi.VariableBase.msm(bases, scalars))
    assert (got == orc.msm_pippenger(bases, scalars, threads=8)).all()
    # only the cancelling part: the whole MSM is the point at infinity
    m = (n // 8) & ~1
    got = jac_affine(mi.VariableBase.msm(bases[q:q + m], scalars[q:q + m]))
    assert not got.any()
