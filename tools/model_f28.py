#!/usr/bin/env python3
"""Big-integer model of the reduced-radix field arithmetic of the bucket accumulation (snarkos_amd/csrc/field28.hpp).

Fq values travel as 14 SIGNED limbs of 28 bits (value = sum l_i 2^(28 i)), so that
  * a limb product fits a signed 64-bit accumulator with room for a whole product-scanning column: ONE
    v_mad_i64_i32 per limb product and no carry instruction (the 32-bit-limb product needs v_mad_u64_u32 +
    v_addc_co_u32 per limb product);
  * additions and subtractions are limb-wise with no carry chain and no modular correction (signed limbs, lazy values).
The Montgomery radix of this domain is R' = 2^392.  The library's data is in Montgomery form for R = 2^384; a loaded
value x R is turned into x R' by reading it 8 bits higher (x R 2^8 < 2^385 fits the 14 limbs un-reduced), and a result
goes back through one product with 2^384 (v R' * 2^384 / R' ... = v 2^-8).

This file models every operation limb for limb with Python integers, asserts the accumulator and limb bounds the HIP
code relies on with worst-case magnitudes, and checks the values (tests/test_f28_model.py runs it).
"""
import random

X = 0x8508C00000000001
R_ORD = X**4 - X**2 + 1
Q = (X - 1) ** 2 * R_ORD // 3 + X
W = 28
MASK = (1 << W) - 1
N = 14
RP = 1 << (W * N)  # R' = 2^392
R32 = 1 << 384
I64_MAX = (1 << 63) - 1

PL = [(Q >> (W * i)) & MASK for i in range(N)]
assert PL[0] == 1  # q = 1 mod 2^32: the Montgomery factor of a column is minus its low limb


def val(l):
    return sum(x << (W * i) for i, x in enumerate(l))


def split(v):
    """canonical split of a non-negative value below 2^392 into 14 limbs"""
    assert 0 <= v < RP
    return [(v >> (W * i)) & MASK for i in range(N)]


def load32(x_mont):
    """a library value (x R mod q, canonical) as x R' (lazy): read 8 bits higher"""
    assert 0 <= x_mont < Q
    return split(x_mont << 8)


def add(a, b):
    return [x + y for x, y in zip(a, b)]


def sub(a, b):
    return [x - y for x, y in zip(a, b)]


def dbl(a):
    return [2 * x for x in a]


def neg(a):
    return [-x for x in a]


def normalise(a):
    """carry pass: limbs 0..12 into [0, 2^28), the top limb keeps the sign"""
    l = list(a)
    for i in range(N - 1):
        c = l[i] >> W  # arithmetic shift
        l[i] &= MASK
        l[i + 1] += c
        assert -(1 << 31) <= l[i + 1] < (1 << 31)
    assert val(l) == val(a)
    return l


def mul(a, b, stats=None):
    """Montgomery product a b / R' mod q: limbs 0..12 of the result in [0, 2^28), the top limb signed; value in
    (-|a b| / R', |a b| / R' + q).  Every accumulator value is checked against the signed 64-bit range."""
    acc = 0
    m = [0] * N
    r = [0] * N
    for k in range(2 * N - 1):
        for i in range(N):
            j = k - i
            if 0 <= j < N:
                acc += a[i] * b[j]
                assert abs(acc) <= I64_MAX
        for i in range(min(k, N)):
            j = k - i
            if 1 <= j < N:
                acc += m[i] * PL[j]
                assert abs(acc) <= I64_MAX
        if k < N:
            lo = acc & MASK
            m[k] = (-lo) & MASK
            acc += m[k]  # p_0 = 1
            assert abs(acc) <= I64_MAX and acc & MASK == 0
            acc >>= W
        else:
            r[k - N] = acc & MASK
            acc >>= W
        if stats is not None:
            stats["max_acc_bits"] = max(stats.get("max_acc_bits", 0), abs(acc).bit_length())
    r[N - 1] = acc
    assert -(1 << 31) <= acc < (1 << 31)
    va, vb, vr = val(a), val(b), val(r)
    assert (vr * RP - va * vb) % Q == 0
    bound = abs(va * vb) // RP + 1
    assert -bound <= vr <= bound + Q
    return r


def sqr(a, stats=None):
    """the squaring takes a_i a_j (i < j) once against the doubled limb: same columns, same value"""
    a2 = dbl(a)
    acc = 0
    m = [0] * N
    r = [0] * N
    for k in range(2 * N - 1):
        for i in range(N):
            j = k - i
            if i < j < N:
                acc += a[i] * a2[j]
            elif i == j:
                acc += a[i] * a[i]
            assert abs(acc) <= I64_MAX
        for i in range(min(k, N)):
            j = k - i
            if 1 <= j < N:
                acc += m[i] * PL[j]
                assert abs(acc) <= I64_MAX
        if k < N:
            lo = acc & MASK
            m[k] = (-lo) & MASK
            acc = (acc + m[k]) >> W
        else:
            r[k - N] = acc & MASK
            acc >>= W
    r[N - 1] = acc
    assert r == mul(a, a)
    return r


ONE = split(RP % Q)          # 1 in this domain
TO32 = split(R32)            # v R' -> v R (= v R' 2^-8): mul(v, 2^384)
assert TO32[13] == 1 << 20 and not any(TO32[:13])


def store32(a):
    """back to the library's canonical Montgomery form; |a| < 256 q (every value the group law leaves: |a| < 8 q)"""
    assert abs(val(a)) < 256 * Q
    r = normalise(mul(a, TO32))
    v = val(r)
    assert -Q < v < 2 * Q
    if v < 0:
        v += Q
    if v >= Q:
        v -= Q
    return v


def is_zero_mod_q(a, k_max):
    """a = 0 mod q for a lazy value in (-k_max q, k_max q): low-limb filter, then exact compare"""
    l = normalise(a)
    v = val(l)
    assert abs(v) < k_max * Q
    quick = ((l[0] + k_max) & MASK) <= 2 * k_max  # k q has low limb k: q = 1 mod 2^28
    exact = v % Q == 0
    assert (not exact) or quick  # the filter never misses
    return exact


# ---------------------------------------------------------------------------------------------------------------
# the XYZZ mixed addition (madd-2008-s) in this arithmetic, with the limb magnitudes of every operand asserted
def limb_mag(a):
    return max(abs(x) for x in a)


def madd(acc, pt, stats):
    """acc = (x, y, zz, zzz): x, y normalised lazy values, zz, zzz product outputs; pt = (x2, y2) loaded coordinates.
    Returns the new accumulator or None for the degenerate cases (the kernel takes its slow path there)."""
    x1, y1, zz, zzz = acc
    x2, y2 = pt
    for t in (x1, y1, zz, zzz, x2, y2):
        assert limb_mag(t[:13]) < (1 << 28) and abs(t[13]) < (1 << 29)
    u2 = mul(x2, zz, stats)
    s2 = mul(y2, zzz, stats)
    p = sub(u2, x1)
    r = sub(s2, y1)
    if is_zero_mod_q(p, 8):
        return None
    pp = sqr(p, stats)
    ppp = mul(p, pp, stats)
    qq = mul(x1, pp, stats)
    x3 = sub(sub(sqr(r, stats), ppp), dbl(qq))
    t = sub(qq, x3)
    assert limb_mag(t) < (1 << 31)
    y3 = sub(mul(r, t, stats), mul(y1, ppp, stats))
    zz3 = mul(zz, pp, stats)
    zzz3 = mul(zzz, ppp, stats)
    x3, y3 = normalise(x3), normalise(y3)
    assert abs(val(x3)) < 6 * Q and abs(val(y3)) < 3 * Q
    return (x3, y3, zz3, zzz3)


def xyzz_add(A, B, stats):
    """full XYZZ addition (add-2008-s); operands: x, y with limbs below 2^28 (normalised or loaded), zz, zzz product
    outputs or loaded; None for the degenerate cases"""
    x1, y1, zz1, zzz1 = A
    x2, y2, zz2, zzz2 = B
    for t in A + B:
        lim(t, 1 << 28)
    u1, u2 = mul(x1, zz2, stats), mul(x2, zz1, stats)
    s1, s2 = mul(y1, zzz2, stats), mul(y2, zzz1, stats)
    p, r = lim(sub(u2, u1), 1 << 28), lim(sub(s2, s1), 1 << 28)
    if is_zero_mod_q(p, 8):
        return None
    pp = sqr(p, stats)
    ppp = mul(p, pp, stats)
    qq = mul(u1, pp, stats)
    x3 = sub(sub(sqr(r, stats), ppp), dbl(qq))
    assert mul_fits(1 << 28, 1 << 31)
    y3 = sub(mul(r, lim(sub(qq, x3), 1 << 31), stats), mul(s1, ppp, stats))
    zz3 = mul(mul(zz1, zz2, stats), pp, stats)
    zzz3 = mul(mul(zzz1, zzz2, stats), ppp, stats)
    return (normalise(x3), normalise(y3), zz3, zzz3)


def xyzz_dbl(A, stats):
    """2 (X, Y, ZZ, ZZZ), a = 0 (dbl-2008-s-1)"""
    x, y, zz, zzz = A
    for t in A:
        lim(t, 1 << 28)
    u = dbl(y)
    v = sqr(u, stats)
    w = mul(u, v, stats)
    s_ = mul(x, v, stats)
    xx = sqr(x, stats)
    m = normalise(add(dbl(xx), xx))
    x3 = sub(sqr(m, stats), dbl(s_))
    y3 = sub(mul(m, lim(sub(s_, x3), 1 << 31), stats), mul(w, y, stats))
    return (normalise(x3), normalise(y3), mul(v, zz, stats), mul(w, zzz, stats))


# ---------------------------------------------------------------------------------------------------------------
# the Jacobian doubling and mixed addition of the fallback's scalar-multiplication chains (glv.hpp) in this arithmetic
def mul_fits(la, lb):
    """a column of 14 products of limbs below la, lb plus the reduction's 13 products and the carry stays in 64 bits"""
    return 14 * la * lb + 13 * (1 << 56) + (1 << 37) < (1 << 63)


def lim(a, bound):
    assert limb_mag(a[:13]) < bound, (limb_mag(a[:13]).bit_length(), bound.bit_length())
    return a


def jac_dbl(P, stats):
    """(X, Y, Z) -> 2 (X, Y, Z), a = 0 (dbl-2009-l); X, Y normalised (limbs < 2^28), Z with limbs < 2^29"""
    X, Y, Z = P
    lim(X, 1 << 28), lim(Y, 1 << 28), lim(Z, 1 << 29)
    A, B = sqr(X, stats), sqr(Y, stats)
    C = sqr(B, stats)
    assert mul_fits(1 << 29, 1 << 29)
    t = sqr(lim(add(X, B), 1 << 29), stats)
    D = normalise(dbl(sub(sub(t, A), C)))          # limbs of the difference below 3 2^28, doubled below 2^31
    E = normalise(add(dbl(A), A))                  # 3 A
    F = sqr(E, stats)
    X3 = normalise(sub(F, dbl(D)))
    assert mul_fits(1 << 28, 1 << 30)
    Y3 = normalise(sub(mul(E, lim(sub(D, X3), 1 << 30), stats), lim([8 * c for c in C], 1 << 31)))
    Z3 = dbl(mul(Y, Z, stats))
    return (X3, Y3, Z3)


def jac_madd(P, pt, stats):
    """(X, Y, Z) + affine (x2, y2) (madd-2007-bl), x2, y2 as loaded; None for the degenerate cases"""
    X1, Y1, Z1 = P
    x2, y2 = pt
    lim(X1, 1 << 28), lim(Y1, 1 << 28), lim(Z1, 1 << 29), lim(x2, 1 << 28), lim(y2, 1 << 28)
    z1z1 = sqr(Z1, stats)
    u2 = mul(x2, z1z1, stats)
    s2 = mul(mul(y2, Z1, stats), z1z1, stats)
    H = lim(sub(u2, X1), 1 << 29)
    if is_zero_mod_q(H, 8):
        return None
    rr = lim(sub(s2, Y1), 1 << 29)                  # r / 2
    HH = sqr(H, stats)
    I = [4 * x for x in HH]                         # limbs below 2^30
    assert mul_fits(1 << 29, 1 << 30)
    J = mul(H, lim(I, 1 << 30), stats)
    V = mul(X1, I, stats)
    X3 = normalise(sub(sub([4 * x for x in sqr(rr, stats)], J), dbl(V)))
    Y3 = normalise(dbl(sub(mul(rr, lim(sub(V, X3), 1 << 29), stats), mul(Y1, J, stats))))
    Z3 = sub(sub(sqr(normalise(add(Z1, H)), stats), z1z1), HH)  # (the sum's limbs reach 2^30: carried before the squaring)
    # Z3 feeds the next doubling, which wants limbs below 2^29
    return (X3, Y3, normalise(Z3))


def self_test(n=60, seed=3):
    rnd = random.Random(seed)
    stats = {}
    # worst-case limb magnitudes a product must take: one operand with limbs below 2^28 and one below 2^31 (r * (qq - x3)),
    # or both below 2^29 (differences of normalised values)
    # (values stay below 2^385, so the top limb -- bits 364 and up -- is small whatever the lower limbs carry)
    top = 1 << 21
    big = [(1 << 31) - 1] * 13 + [top]
    sm = [MASK] * 13 + [top]
    mid = [(1 << 29) - 1] * 13 + [top]
    for a, b in ((sm, big), (big, sm), ([-x for x in sm], big), (mid, [-x for x in mid]), (mid, mid)):
        mul(a, b, stats)
    sqr(mid)
    sqr([-x for x in mid])
    rinv = pow(RP, -1, Q)
    for _ in range(n):
        x, y = rnd.randrange(Q), rnd.randrange(Q)
        a, b = load32(x), load32(y)
        c = mul(a, b, stats)
        assert val(c) % Q == (x << 8) * (y << 8) * rinv % Q
        assert store32(c) == x * y * pow(R32, -1, Q) % Q  # the library's Montgomery product of x and y
        assert store32(a) == x
        c2, c3 = mul(a, a, stats), mul(b, c, stats)
        d = sub(sub(c, c2), dbl(c3))  # lazy combination of product outputs, as x3 in the group law: |d| < 5 q
        assert store32(d) == (val(c) - val(c2) - 2 * val(c3)) * pow(256, -1, Q) % Q
        assert is_zero_mod_q(sub(c, c), 4) and (not is_zero_mod_q(sub(c, c2), 4) or val(c) % Q == val(c2) % Q)
        assert is_zero_mod_q(sub(c, add(c, split(Q))), 4) and is_zero_mod_q(sub(add(c, split(3 * Q)), c), 4)
    # a chain of mixed additions on the curve: values against the affine group law
    from_mont = pow(R32, -1, Q)

    def aff_add(p1, p2):
        (x1, y1), (x2, y2) = p1, p2
        lam = (y2 - y1) * pow(x2 - x1, -1, Q) % Q
        x3 = (lam * lam - x1 - x2) % Q
        return (x3, (lam * (x1 - x3) - y1) % Q)

    def rand_point():
        while True:
            x = rnd.randrange(Q)
            y2 = (x**3 + 1) % Q
            y = pow(y2, (Q + 1) // 4, Q) if Q % 4 == 3 else None
            if y is None:  # q = 1 mod 4: Tonelli-Shanks through a library-free route: try until a square shows by Euler
                if pow(y2, (Q - 1) // 2, Q) != 1:
                    continue
                # q - 1 = 2^46 * odd: plain Tonelli-Shanks
                s, t = 0, Q - 1
                while t % 2 == 0:
                    s += 1
                    t //= 2
                z = 2
                while pow(z, (Q - 1) // 2, Q) != Q - 1:
                    z += 1
                c, r_, tt, m_ = pow(z, t, Q), pow(y2, (t + 1) // 2, Q), pow(y2, t, Q), s
                while tt != 1:
                    i2, t2 = 0, tt
                    while t2 != 1:
                        t2 = t2 * t2 % Q
                        i2 += 1
                    b_ = pow(c, 1 << (m_ - i2 - 1), Q)
                    r_, c, tt, m_ = r_ * b_ % Q, b_ * b_ % Q, tt * b_ * b_ % Q, i2
                y = r_
            if y * y % Q == y2:
                return (x, y)

    pts = [rand_point() for _ in range(12)]
    accp = pts[0]
    # the first point of a chunk enters through a product with one: loaded coordinates are lazy values up to 256 q, the
    # accumulator's x and y must stay small for the zero test of p
    acc = (mul(load32(accp[0] * R32 % Q), ONE), mul(load32(accp[1] * R32 % Q), ONE), ONE, ONE)
    for ptn in pts[1:]:
        pt28 = (load32(ptn[0] * R32 % Q), load32(ptn[1] * R32 % Q))
        acc = madd(acc, pt28, stats)
        assert acc is not None
        accp = aff_add(accp, ptn)
        zz, zzz = store32(acc[2]) * from_mont % Q, store32(acc[3]) * from_mont % Q
        xa = store32(acc[0]) * from_mont * pow(zz, -1, Q) % Q
        ya = store32(acc[1]) * from_mont * pow(zzz, -1, Q) % Q
        assert (xa, ya) == accp
    # full XYZZ additions and doublings: a small tree of sums against the affine group law
    def to_aff(A):
        zz, zzz = store32(A[2]) * from_mont % Q, store32(A[3]) * from_mont % Q
        return (store32(A[0]) * from_mont * pow(zz, -1, Q) % Q, store32(A[1]) * from_mont * pow(zzz, -1, Q) % Q)

    def lift(pt):
        return (load32(pt[0] * R32 % Q), load32(pt[1] * R32 % Q), load32(R32 % Q), load32(R32 % Q))

    s01 = xyzz_add(lift(pts[0]), lift(pts[1]), stats)
    s23 = xyzz_add(lift(pts[2]), lift(pts[3]), stats)
    assert to_aff(s01) == aff_add(pts[0], pts[1])
    s0123 = xyzz_add(s01, s23, stats)
    assert to_aff(s0123) == aff_add(aff_add(pts[0], pts[1]), aff_add(pts[2], pts[3]))
    assert xyzz_add(s01, s01, stats) is None and xyzz_add(s01, lift(aff_add(pts[0], pts[1])), stats) is None
    d = xyzz_dbl(s0123, stats)
    s_aff = to_aff(s0123)
    lam = 3 * s_aff[0] * s_aff[0] * pow(2 * s_aff[1], -1, Q) % Q
    dx = (lam * lam - 2 * s_aff[0]) % Q
    assert to_aff(d) == (dx, (lam * (s_aff[0] - dx) - s_aff[1]) % Q)
    assert to_aff(xyzz_dbl(lift(pts[4]), stats))[0] == (pow(3 * pts[4][0] ** 2 * pow(2 * pts[4][1], -1, Q), 2, Q) - 2 * pts[4][0]) % Q
    # a Jacobian double-and-add chain against the affine group law
    def aff_dbl(p1):
        x1, y1 = p1
        lam = 3 * x1 * x1 * pow(2 * y1, -1, Q) % Q
        x3 = (lam * lam - 2 * x1) % Q
        return (x3, (lam * (x1 - x3) - y1) % Q)

    base = pts[3]
    b28 = (load32(base[0] * R32 % Q), load32(base[1] * R32 % Q))
    J0 = (mul(b28[0], ONE), mul(b28[1], ONE), ONE)
    ref = base
    scalar = 0b110101101110010111
    for bit in bin(scalar)[3:]:
        J0 = jac_dbl(J0, stats)
        ref = aff_dbl(ref)
        if bit == "1":
            J0 = jac_madd(J0, b28, stats)
            ref = aff_add(ref, base)
        z = store32(J0[2]) * from_mont % Q
        assert (store32(J0[0]) * from_mont * pow(z * z, -1, Q) % Q, store32(J0[1]) * from_mont * pow(z * z * z, -1, Q) % Q) == ref
    assert jac_madd((mul(b28[0], ONE), mul(b28[1], ONE), ONE), b28, stats) is None
    mul([(1 << 29) - 1] * 13 + [1 << 21], [-((1 << 30) - 1)] * 13 + [1 << 21], stats)
    # the degenerate cases are flagged
    p0 = pts[0]
    l0 = (load32(p0[0] * R32 % Q), load32(p0[1] * R32 % Q))
    a0 = (mul(l0[0], ONE), mul(l0[1], ONE), ONE, ONE)
    assert madd(a0, l0, stats) is None
    assert madd(a0, (l0[0], load32((Q - p0[1]) * R32 % Q)), stats) is None
    return stats


if __name__ == "__main__":
    print(self_test())
