"""-m gpu: prover-side reuse of the MSM / NTT kernels: KZG10::commit and the polynomial product,
bit-exact against the CPU oracle, tied to the verifier side by an end-to-end commit -> open -> check."""
import numpy as np
import pytest

import pyref as P
from helpers import jac_affine, rng_fr_mont, toy_srs

pytestmark = pytest.mark.gpu

TAU = 0x1234567890ABCDEF1122334455667788 % P.R


@pytest.fixture(scope="module")
def srs(orc, mi):
    powers, vk = toy_srs(5200, TAU)
    rb = mi.ResidentBases(powers)
    yield powers, vk, rb
    rb.release()


@pytest.mark.parametrize("n", [1, 2, 100, 5000])
def test_commit_matches_oracle_and_p_of_tau(orc, mi, srs, n):
    powers, _, rb = srs
    coeffs = rng_fr_mont(900 + n, n)
    got = jac_affine(rb.commit(coeffs))
    assert (got == orc.kzg_commit(powers[:n], coeffs, threads=8)).all()
    # commitment to p is p(tau) * G: an identity neither implementation's MSM knows about
    p_tau = orc.fr_to_ints(orc.poly_eval(coeffs, orc.fr_from_ints([TAU])[0]))[0]
    assert (got == orc.g1_mul(orc.g1_generator(), p_tau)).all()


def test_commit_edge_cases(orc, mi, srs):
    import torch
    powers, _, rb = srs
    assert not jac_affine(rb.commit(np.zeros((0, 4), dtype=np.uint64))).any()          # empty polynomial
    assert not jac_affine(rb.commit(np.zeros((17, 4), dtype=np.uint64))).any()         # zero polynomial
    coeffs = rng_fr_mont(77, 300)
    coeffs[5] = 0
    coeffs[9] = orc.fr_from_ints([P.R - 1])[0]
    want = orc.kzg_commit(powers[40:340], coeffs, threads=8)
    assert (jac_affine(rb.commit(coeffs, offset=40)) == want).all()                    # shifted powers
    dev = torch.from_numpy(coeffs.view(np.int64)).cuda()
    assert (jac_affine(rb.commit(dev, offset=40)) == want).all()                       # coefficients already in HBM
    from snarkos_amd.lib import MiError
    with pytest.raises(MiError):
        rb.commit(rng_fr_mont(1, 10), offset=5195)                                     # runs past the powers


def test_commit_is_linear_at_2_16(orc, mi):
    n = 1 << 16
    powers, _ = toy_srs(n, TAU)
    rb = mi.ResidentBases(powers)
    try:
        a, b = rng_fr_mont(11, n), rng_fr_mont(12, n)
        ai, bi = orc.fr_to_ints(a), orc.fr_to_ints(b)
        s = orc.fr_from_ints([(x + y) % P.R for x, y in zip(ai, bi)])
        ca, cb, cs = (jac_affine(rb.commit(x)) for x in (a, b, s))
        assert (orc.g1_add_affine(ca, cb) == cs).all()
        assert (ca == orc.kzg_commit(powers, a, threads=8)).all()
    finally:
        rb.release()


@pytest.mark.parametrize("na,nb", [(1, 1), (7, 5), (1000, 1), (300, 257), (1024, 1025)])
def test_poly_mul_matches_schoolbook(orc, mi, na, nb):
    a, b = rng_fr_mont(3000 + na, na), rng_fr_mont(4000 + nb, nb)
    got = mi.DensePolynomial.mul(a, b)
    assert got.shape == (na + nb - 1, 4)
    assert (got == orc.poly_mul_schoolbook(a, b)).all()


def test_poly_mul_large_by_evaluation(orc, mi):
    """2^17-coefficient operands: (a b)(z) = a(z) b(z) at a random z, and the top coefficient is the product of the tops"""
    n = 1 << 17
    a, b = rng_fr_mont(51, n), rng_fr_mont(52, n - 3)
    got = mi.DensePolynomial.mul(a, b)
    z = orc.fr_from_ints([0x5EED5EED5EED5EED5EED5EED5EED % P.R])[0]
    ev = lambda c: orc.fr_to_ints(orc.poly_eval(c, z))[0]
    assert ev(got) == ev(a) * ev(b) % P.R
    assert orc.fr_to_ints(got[-1:])[0] == orc.fr_to_ints(a[-1:])[0] * orc.fr_to_ints(b[-1:])[0] % P.R


def test_commit_open_check_end_to_end(orc, mi, srs):
    """prover side on the GPU (commit p, commit the witness polynomial), verifier side on the GPU (KZG10 check):
    the pairing equation only holds if MSM, the conversion of coefficients and the check agree with the algebra"""
    powers, vk, rb = srs
    p = rng_fr_mont(7001, 4096)
    z = orc.fr_from_ints([0xC0FFEE1234567 % P.R])[0]
    q, value = orc.poly_witness(p, z)
    assert (value == orc.poly_eval(p, z)).all()
    comm, wit = jac_affine(rb.commit(p)), jac_affine(rb.commit(q))
    assert mi.KZG10.check(vk, comm, z, value, wit)
    assert orc.kzg_batch_check(vk, comm[None], z[None], value[None], wit[None], np.frombuffer((1).to_bytes(16, "little"), dtype=np.uint8))
    bad = orc.fr_from_ints([(orc.fr_to_ints(value)[0] + 1) % P.R])[0]
    assert not mi.KZG10.check(vk, comm, z, bad, wit)
    # a product polynomial commits to the product of evaluations at tau
    f, g = rng_fr_mont(7002, 1500), rng_fr_mont(7003, 2000)
    fg = mi.DensePolynomial.mul(f, g)
    t = orc.fr_from_ints([TAU])[0]
    want = orc.fr_to_ints(orc.poly_eval(f, t))[0] * orc.fr_to_ints(orc.poly_eval(g, t))[0] % P.R
    assert (jac_affine(rb.commit(fg)) == orc.g1_mul(orc.g1_generator(), want)).all()


@pytest.mark.parametrize("n", [1, 2, 9, 2048, 2049, 5000])
def test_open_matches_oracle(orc, mi, srs, n):
    """KZG10::open: value and witness commitment against the oracle's synthetic division + Pippenger"""
    powers, vk, rb = srs
    p = rng_fr_mont(8100 + n, n)
    z = orc.fr_from_ints([(0xDEADBEEF00 + n) % P.R])[0]
    value, wit = rb.open(p, z)
    q, want_v = orc.poly_witness(p, z)
    assert (value == want_v).all()
    want_w = orc.kzg_commit(powers[:max(n - 1, 1)], q, threads=8) if n > 1 else np.zeros(12, dtype=np.uint64)
    assert (jac_affine(wit) == want_w).all()
    if n > 1:
        assert mi.KZG10.check(vk, jac_affine(rb.commit(p)), z, value, jac_affine(wit))


def test_open_edge_cases_and_large(orc, mi):
    n = (1 << 16) + 7   # several scan blocks, ragged tail
    powers, vk = toy_srs(n, TAU)
    rb = mi.ResidentBases(powers)
    try:
        p = rng_fr_mont(8200, n)
        p[n - 1] = 0                                     # leading zero coefficient
        for zi in (0, 1, 0x1234567 % P.R):               # z = 0: value = c_0, q = shift; z = 1: value = sum
            z = orc.fr_from_ints([zi])[0]
            value, wit = rb.open(p, z)
            q, want_v = orc.poly_witness(p, z)
            assert (value == want_v).all()
            assert (jac_affine(wit) == orc.kzg_commit(powers[:n - 1], q, threads=8)).all()
        assert mi.KZG10.check(vk, jac_affine(rb.commit(p)), z, value, jac_affine(wit))
        value, wit = rb.open(np.zeros((0, 4), dtype=np.uint64), z)
        assert not value.any() and not jac_affine(wit).any()
    finally:
        rb.release()


def test_kzg_hiding_vector_and_end_to_end(orc, mi):
    """row A5 with hiding: the pyref-generated vector through mi_kzg_batch_check_hiding, then commit_hiding -> open_hiding ->
    batch_check_hiding on the device with gamma_g powers as a second resident table; a wrong random_v is refused"""
    import golden_util as G
    d = G.load("kzg_hiding.json")
    vk = orc.kzg_vk(1)
    vk[36:60] = orc.g2_from_ints(G.g2_pt(d["tau_h"]))
    gg = G.g1_arr([d["gamma_g"]])[0]
    cs, ws = G.g1_arr(d["comms"]), G.g1_arr(d["witnesses"])
    zs, vs, rnd = G.fr_arr(d["points"]), G.fr_arr(d["values"]), G.rand_arr(d["randomizers"])
    assert mi.KZG10.batch_check(vk, cs, zs, vs, ws, rnd, gamma_g=gg, random_v=G.fr_arr(d["random_v"]))
    assert not mi.KZG10.batch_check(vk, cs, zs, vs, ws, rnd, gamma_g=gg, random_v=G.fr_arr(d["bad_random_v"]))
    assert not mi.KZG10.batch_check(vk, cs, zs, vs, ws, rnd)
    # end to end with a toy SRS: gamma_g = gamma G, powers_of_beta_times_gamma_g[i] = gamma tau^i G
    gamma = 0xC0FFEE1234567 % P.R
    n, m = 700, 3
    powers, vk = toy_srs(n, TAU)
    gpw, acc = [], gamma
    for _ in range(m):
        gpw.append(acc)
        acc = acc * TAU % P.R
    gpowers = orc.fixed_base_batch(orc.fr_from_ints(gpw))
    rb, rg = mi.ResidentBases(powers), mi.ResidentBases(gpowers)
    try:
        coeffs, rho = rng_fr_mont(4001, n), rng_fr_mont(4002, m)
        c = jac_affine(rb.commit_hiding(rg, coeffs, rho))
        want = orc.g1_add_affine(orc.kzg_commit(powers, coeffs, threads=8), orc.kzg_commit(gpowers, rho, threads=2))
        assert (c == want).all()
        z = orc.fr_from_ints([0x1337])[0]
        v, rv, w = rb.open_hiding(rg, coeffs, rho, z)
        assert (v == orc.poly_eval(coeffs, z)).all() and (rv == orc.poly_eval(rho, z)).all()
        w = jac_affine(w)
        q, _ = orc.poly_witness(coeffs, z)
        qr, _ = orc.poly_witness(rho, z)
        assert (w == orc.g1_add_affine(orc.kzg_commit(powers[:n - 1], q, threads=8), orc.kzg_commit(gpowers[:m - 1], qr, threads=2))).all()
        one = np.zeros(16, dtype=np.uint8); one[0] = 1
        assert mi.KZG10.batch_check(vk, [c], [z], [v], [w], one, gamma_g=gpowers[0], random_v=[rv])
        assert orc.kzg_batch_check_hiding(vk, gpowers[0], np.stack([c]), np.stack([z]), np.stack([v]), np.stack([w]), np.stack([rv]), one)
        bad = rv.copy(); bad[0] ^= np.uint64(1)
        assert not mi.KZG10.batch_check(vk, [c], [z], [v], [w], one, gamma_g=gpowers[0], random_v=[bad])
        assert not mi.KZG10.batch_check(vk, [c], [z], [v], [w], one)  # without the gamma_g term the hiding commitment does not open
    finally:
        rb.release()
        rg.release()


def test_commit_lagrange_equals_commit_of_the_interpolant(orc, mi):
    """KZG10::commit_lagrange: evaluations over the domain against L_i(tau) G give the commitment of the interpolating
    polynomial (ifft of the evaluations against the monomial powers)"""
    n = 256
    w = P.omega(8)
    # L_i(tau) = (tau^n - 1) w^i / (n (tau - w^i))
    zn = (pow(TAU, n, P.R) - 1) % P.R
    lag = [zn * pow(w, i, P.R) % P.R * P.inv(n * (TAU - pow(w, i, P.R)) % P.R, P.R) % P.R for i in range(n)]
    lbases = orc.fixed_base_batch(orc.fr_from_ints(lag))
    powers, _ = toy_srs(n, TAU)
    rl, rp = mi.ResidentBases(lbases), mi.ResidentBases(powers)
    try:
        evals = rng_fr_mont(5001, n)
        coeffs = mi.EvaluationDomain(n).ifft(evals)
        assert (jac_affine(rl.commit_lagrange(evals)) == jac_affine(rp.commit(coeffs))).all()
        assert (jac_affine(rl.commit_lagrange(evals)) == orc.kzg_commit(lbases, evals, threads=4)).all()
    finally:
        rl.release()
        rp.release()


@pytest.mark.parametrize("log_m,log_n", [(10, 10), (12, 10), (14, 9), (16, 16)])
def test_divide_by_vanishing_poly_on_coset(orc, mi, log_m, log_n):
    """EvaluationDomain::divide_by_vanishing_poly_on_coset_in_place: p = q (X^N - 1) with deg q < M - N; coset_fft(p) over
    the size-M domain, divided on the coset, coset_ifft -> q.  The quotient exists only when M > N; at M = N the pointwise
    factor is the constant 1 / (g^N - 1), checked directly."""
    import torch
    M, N = 1 << log_m, 1 << log_n
    d = mi.EvaluationDomain(M)
    if M == N:
        a = rng_fr_mont(6000 + log_m, M)
        got = orc.fr_to_ints(d.divide_by_vanishing_poly_on_coset(a, N))
        c = P.inv((pow(22, N, P.R) - 1) % P.R, P.R)
        assert got == [x * c % P.R for x in orc.fr_to_ints(a)]
        return
    q = rng_fr_mont(6100 + log_m, M - N)
    p = np.zeros((M, 4), dtype=np.uint64)
    qi = orc.fr_to_ints(q)
    pi = [0] * M
    for i, x in enumerate(qi):
        pi[i] = (pi[i] - x) % P.R
        pi[i + N] = (pi[i + N] + x) % P.R
    p[:] = orc.fr_from_ints(pi)
    ev = d.coset_fft(p)
    quot = d.coset_ifft(d.divide_by_vanishing_poly_on_coset(ev, N))
    assert (quot[:M - N] == q).all() and not quot[M - N:].any()
    t = torch.from_numpy(ev.view(np.int64)).cuda()
    d.divide_by_vanishing_poly_on_coset(t, N)
    d.coset_ifft(t)
    assert (t.cpu().numpy().view(np.uint64)[:M - N] == q).all()
