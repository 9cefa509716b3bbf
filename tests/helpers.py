"""Seeded input builders shared by the parity tests (oracle side only; test infrastructure)."""
import numpy as np

import orc_py as O
import pyref as P


def rng_scalars(seed, n, bits=253):
    """canonical scalars < r as (n,4) uint64"""
    g = np.random.default_rng(seed)
    a = g.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + g.integers(0, 2, size=(n, 4), dtype=np.uint64)
    if bits <= 192:
        a[:, 3] = 0
    if bits <= 128:
        a[:, 2] = 0
    a[:, 3] &= np.uint64((1 << 60) - 1)  # < 2^252 < r
    return np.ascontiguousarray(a)


def rng_points(seed, n):
    """n distinct in-subgroup points k_i * G (Montgomery affine), via the oracle's fixed-base table"""
    k = rng_scalars(seed ^ 0x5eed, n)
    kk = np.zeros_like(k)
    for i in range(n):  # to Montgomery: the fixed-base helper takes Montgomery Fr
        kk[i] = O.fr_from_ints([O.limbs_to_int(k[i])])[0]
    return O.fixed_base_batch(kk)


def rng_fr_mont(seed, n):
    """n Montgomery Fr elements (any 256-bit pattern < r is a valid Montgomery representative)"""
    return rng_scalars(seed, n)


def jac_affine(j):
    """library Jacobian output (normalised) -> (12,) affine array, zeros for infinity"""
    j = np.asarray(j, dtype=np.uint64).reshape(18)
    if not j[12:].any():
        return np.zeros(12, dtype=np.uint64)
    one = np.array([int(x) for x in O.int_to_limbs(P.FQ_MONT_R, 6)], dtype=np.uint64)
    assert (j[12:] == one).all(), "library outputs must be normalised (z = 1)"
    return j[:12].copy()


def toy_srs(n, tau):
    """powers_of_beta_g[i] = tau^i * G for i < n ((n, 12) Montgomery affine) and the matching kzg vk (toy setup: tau known)"""
    pw, acc = [], 1
    for _ in range(n):
        pw.append(acc)
        acc = acc * tau % P.R
    return O.fixed_base_batch(O.fr_from_ints(pw)), O.kzg_vk(tau)
