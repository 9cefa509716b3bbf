"""-m gpu: pairing and KZG10::batch_check parity against the CPU oracle."""
import numpy as np
import pytest

import pyref as P

pytestmark = pytest.mark.gpu


def test_pairing_value_matches_oracle(orc, mi):
    p = orc.g1_mul(orc.g1_generator(), 0x1234567)
    q = orc.g2_mul(orc.g2_generator(), 0x7654321)
    assert (mi.Bls12_377.pairing(p, q) == orc.pairing(p, q)).all()


def test_pairing_value_golden_vector_and_random_pairs(orc, mi):
    """the pairing VALUE (cube of the reduced ate pairing, every coefficient): the pyref-generated golden vector
    tests/golden/pairing.json through the HIP library, a handful of random (P, Q) against the C oracle, bilinearity in
    both arguments on the device's own values, and the points at infinity"""
    import golden_util as G
    d = G.load("pairing.json")
    p, q = orc.g1_from_ints(G.pt(d["p"])), orc.g2_from_ints(G.g2_pt(d["q"]))
    got = mi.Bls12_377.pairing(p, q)
    assert orc.fq12_to_power_basis(got) == [G.h(c) for c in d["e_cubed_power_basis"]]
    rng = P.SplitMix64(0xE0E0)
    g, h = orc.g1_generator(), orc.g2_generator()
    for _ in range(6):
        a, b = rng.below(P.R, 5), rng.below(P.R, 5)
        pa, qb = orc.g1_mul(g, a), orc.g2_mul(h, b)
        e = mi.Bls12_377.pairing(pa, qb)
        assert (e == orc.pairing(pa, qb)).all()
        # e(aG, bH) == e(abG, H) == e(G, abH): three different Miller loops, one value
        ab = a * b % P.R
        assert (mi.Bls12_377.pairing(orc.g1_mul(g, ab), h) == e).all() and (mi.Bls12_377.pairing(g, orc.g2_mul(h, ab)) == e).all()
    one = mi.Bls12_377.pairing(np.zeros(12, dtype=np.uint64), h)
    assert (one == mi.Bls12_377.pairing(g, np.zeros(24, dtype=np.uint64))).all()
    assert orc.fq12_to_power_basis(one) == [1] + [0] * 11


def test_pairing_product(orc, mi):
    g, h = orc.g1_generator(), orc.g2_generator()
    a, b = 1234577, 99991
    # e(aG, bH) * e(-abG, H) == 1
    ab_neg = orc.g1_mul(g, (P.R - a * b % P.R) % P.R)
    ps = np.stack([orc.g1_mul(g, a), ab_neg])
    qs = np.stack([orc.g2_mul(h, b), h])
    assert mi.Bls12_377.product_of_pairings_is_one(ps, qs)
    qs2 = np.stack([orc.g2_mul(h, b + 1), h])
    assert not mi.Bls12_377.product_of_pairings_is_one(ps, qs2)
    # infinity on either side contributes 1
    ps3 = np.stack([np.zeros(12, dtype=np.uint64), ab_neg]); qs3 = np.stack([h, np.zeros(24, dtype=np.uint64)])
    assert mi.Bls12_377.product_of_pairings_is_one(ps3, qs3)


@pytest.mark.parametrize("n", [1, 3, 4])
def test_pairing_product_any_number_of_pairs(orc, mi, n):
    """prod_i e(a_i G, b_i H) * e(-(sum a_i b_i) G, H) == 1 over n + 1 <= 4 pairs (couples of Miller loops multiplied),
    the single-pair loop, and the all-infinity product"""
    g, h = orc.g1_generator(), orc.g2_generator()
    ab = [(1000003 + 17 * i, 777767 + 31 * i) for i in range(n)]
    if n < 4:
        total = sum(a * b for a, b in ab) % P.R
        ps = np.stack([orc.g1_mul(g, a) for a, _ in ab] + [orc.g1_mul(g, (P.R - total) % P.R)])
        qs = np.stack([orc.g2_mul(h, b) for _, b in ab] + [h])
    else:  # four pairs in all: the last one closes the product
        total = sum(a * b for a, b in ab[:3]) % P.R
        ps = np.stack([orc.g1_mul(g, a) for a, _ in ab[:3]] + [orc.g1_mul(g, (P.R - total) % P.R)])
        qs = np.stack([orc.g2_mul(h, b) for _, b in ab[:3]] + [h])
    assert mi.Bls12_377.product_of_pairings_is_one(ps, qs)
    qs_bad = qs.copy(); qs_bad[0] = orc.g2_mul(h, ab[0][1] + 1)
    assert not mi.Bls12_377.product_of_pairings_is_one(ps, qs_bad)
    zero_p = np.zeros_like(ps)
    assert mi.Bls12_377.product_of_pairings_is_one(zero_p, qs)      # every pair at infinity: empty product


def _kzg_instance(orc, n, seed):
    rng = P.SplitMix64(seed)
    tau = rng.below(P.R, 5)
    vk = orc.kzg_vk(tau)
    polys = [[rng.below(P.R, 5) for _ in range(4)] for _ in range(n)]
    ev = lambda c, z: sum(ci * pow(z, i, P.R) for i, ci in enumerate(c)) % P.R
    zs = [rng.below(P.R, 5) for _ in range(n)]
    vs = [ev(c, z) for c, z in zip(polys, zs)]
    cs = orc.fixed_base_batch(orc.fr_from_ints([ev(c, tau) for c in polys]))
    ws = orc.fixed_base_batch(orc.fr_from_ints([(ev(c, tau) - v) * P.inv((tau - z) % P.R, P.R) % P.R for c, z, v in zip(polys, zs, vs)]))
    rnd = np.frombuffer(b"".join(rng.below(1 << 128, 2).to_bytes(16, "little") for _ in range(n)), dtype=np.uint8).copy()
    return vk, cs, orc.fr_from_ints(zs), vs, ws, rnd


@pytest.mark.parametrize("n", [1, 3, 40])
def test_kzg_batch_check(orc, mi, n):
    vk, cs, zs, vs, ws, rnd = _kzg_instance(orc, n, 500 + n)
    vals = orc.fr_from_ints(vs)
    assert orc.kzg_batch_check(vk, cs, zs, vals, ws, rnd)
    assert mi.KZG10.batch_check(vk, cs, zs, vals, ws, rnd)
    bad = list(vs); bad[n // 2] = (bad[n // 2] + 1) % P.R
    badv = orc.fr_from_ints(bad)
    assert not orc.kzg_batch_check(vk, cs, zs, badv, ws, rnd)
    assert not mi.KZG10.batch_check(vk, cs, zs, badv, ws, rnd)
    ws2 = ws.copy(); ws2[0] = orc.g1_generator()
    assert not mi.KZG10.batch_check(vk, cs, zs, vals, ws2, rnd)
    assert mi.KZG10.check(vk, cs[0], zs[0], vals[0], ws[0])
