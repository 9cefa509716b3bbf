#!/usr/bin/env python3
"""Big-integer model of the pairing kernel's Fq12 rounds (snarkos_amd/csrc/pairing_wg.hpp: the general round of round 2
and the cyclotomic squaring round of round 3).

The round computes, for every output coefficient k of C = A * B in Fq[w]/(w^12 + 5),
    c_k = ( sum_{s=0..11} a_((k-s) mod 12) * b'_s ) * R^-1  mod q,   b'_s = b_s (s <= k) or b5_s = 10 q - 5 b_s (s > k),
on a 16-lane row: lane (j = L >> 2, g = L & 3) multiplies the three products s = 3g .. 3g+2 by limbs 3j .. 3j+2 of b'_s
(a 12 x 3-limb product-scanning pass, sums unreduced), the four lanes of a quad add up (same limb offset 3j), every lane
folds four words of its quad's sum with the constants D[m] = 2^(32 m - 384) mod q (m < 12; a plain power of two for
m >= 12: the Montgomery reduction distributed over the lanes), the row adds up, and one Barrett step with a two-word
quotient leaves c_k in [0, 2q).  This file checks the arithmetic identities and every bound the HIP code relies on
(tests/test_pairing_round_model.py runs it), with worst-case and random inputs.
"""
import random

X = 0x8508C00000000001
R_ORD = X**4 - X**2 + 1
Q = (X - 1) ** 2 * R_ORD // 3 + X
RM = 1 << 384  # Montgomery radix of Fq
W = 1 << 32


def limbs(v, n):
    return [(v >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def unlimbs(l):
    return sum(x << (32 * i) for i, x in enumerate(l))


# fold constants: a word at limb position m of the unreduced sum T contributes word * 2^(32 m) * R^-1
D = [pow(2, 32 * m, Q) * pow(RM, -1, Q) % Q if m < 12 else (1 << (32 * (m - 12))) % Q for m in range(25)]
MU = (1 << 416) // Q  # Barrett constant for values below 2^416, quotient estimated from the bits above 2^352


def twin5(c):
    """the wrapped operand 10 q - 5 c for c in [0, 2q]"""
    assert 0 <= c <= 2 * Q
    t = 10 * Q - 5 * c
    assert 0 <= t < RM  # fits 12 limbs
    return t


def barrett(v):
    assert 0 <= v < (1 << 416)
    vtop = v >> 352
    assert vtop < (1 << 63)
    qhat = (vtop * MU) >> 64
    assert qhat < (1 << 39)
    c = v - qhat * Q
    assert 0 <= c < 2 * Q, (c.bit_length(), Q.bit_length())
    # the device computes the difference modulo 2^384 only
    assert c == (v - (qhat * Q)) % RM
    return c


def round_row(A, B, B5, k, stats=None):
    """one output coefficient; A, B in [0, 2q], B5 = twin5(B)"""
    P = {}
    for L in range(16):
        j, g = L >> 2, L & 3
        acc = 0
        for t in range(3):
            s = 3 * g + t
            a = A[(k - s) % 12]
            b = (B[s] if s <= k else B5[s])
            chunk = (b >> (96 * j)) & ((1 << 96) - 1)
            acc += a * chunk
        assert acc < (1 << (32 * 16))
        P[L] = acc
    F = 0
    T = 0
    for j in range(4):
        Qj = sum(P[4 * j + g] for g in range(4))
        assert Qj < (1 << (32 * 16))
        T += Qj << (96 * j)
        ql = limbs(Qj, 16)
        for g in range(4):
            f = sum(ql[g + 4 * t] * D[3 * j + g + 4 * t] for t in range(4))
            assert f < (1 << (32 * 13))
            F += f
    assert F < (1 << 416)
    if stats is not None:
        stats["max_F_bits"] = max(stats.get("max_F_bits", 0), F.bit_length())
    c = barrett(F)
    assert c % Q == T * pow(RM, -1, Q) % Q
    return c


def fq12_mul_ref(A, B):
    """schoolbook product in Fq[w]/(w^12 + 5), canonical coefficients"""
    out = [0] * 12
    for i in range(12):
        for j in range(12):
            v = A[i] * B[j]
            if i + j >= 12:
                out[i + j - 12] -= 5 * v
            else:
                out[i + j] += v
    return [x % Q for x in out]


def check(A, B, stats=None):
    B5 = [twin5(b) for b in B]
    C = [round_row(A, B, B5, k, stats) for k in range(12)]
    rinv = pow(RM, -1, Q)
    want = [x * rinv % Q for x in fq12_mul_ref([a % Q for a in A], [b % Q for b in B])]
    assert [c % Q for c in C] == want
    for c in C:
        twin5(c)  # the output's twin fits too
    return C


# ---- the cyclotomic squaring round (pb_csqr_impl): one term per lane g of the quad
def csqr_table():
    """row k -> four terms (a, b, twin, multiplier): derived here from the Granger-Scott formulas in the basis W = w^3,
    compared with PW_CSQR_TABLE of pairing_wg.hpp by tests/test_pairing_round_model.py"""
    def sq_terms(i):  # [X^2]_i over W^4 = -5 as (m, n, wrapped, multiplicity) on the W-coefficients x_m x_n
        out = []
        for m in range(4):
            for n in range(m, 4):
                if (m + n) % 4 == i:
                    out.append((m, n, m + n >= 4, 1 if m == n else 2))
        return out
    rows = {}
    for i in range(4):
        # A-row k = 3i: 3 [A^2]_i - 2 s (-1)^i c_k, x_m = c_(3m)
        rows[3 * i] = [(3 * m, 3 * n, int(wr), 3 * mu) for m, n, wr, mu in sq_terms(i)] + [None] * (3 - len(sq_terms(i))) + \
                      [(3 * i, 12, 1 if i % 2 == 0 else 0, 2)]
        # B-row k = 3i + 2: 3 [B^2]_i - 2 s (-1)^i c_k, x_m = c_(3m+1)
        rows[3 * i + 2] = [(3 * m + 1, 3 * n + 1, int(wr), 3 * mu) for m, n, wr, mu in sq_terms(i)] + [None] * (3 - len(sq_terms(i))) + \
                          [(3 * i + 2, 12, 1 if i % 2 == 0 else 0, 2)]
        # C-row k = 3i + 1: 3 [W C^2]_i + 2 s (-1)^i c_k, x_m = c_(3m+2); [W C^2]_0 = -5 [C^2]_3, [W C^2]_i = [C^2]_(i-1)
        src = sq_terms(3) if i == 0 else sq_terms(i - 1)
        terms = [(3 * m + 2, 3 * n + 2, 1 if i == 0 else int(wr), 3 * mu) for m, n, wr, mu in src]
        rows[3 * i + 1] = terms + [None] * (3 - len(terms)) + [(3 * i + 1, 12, 0 if i % 2 == 0 else 1, 2)]
    rows[12] = [(12, 12, 0, 1), None, None, None]
    return rows


def csqr_row(C, k, table, stats=None):
    """one output coefficient of the squaring of C (13 entries in [0, 2q]: twelve coefficients and the scale)"""
    T5 = [twin5(c) for c in C[:12]] + [2 * Q - C[12]]
    P = {}
    for L in range(16):
        j, g = L >> 2, L & 3
        term = table[k][g]
        acc = 0
        if term is not None:
            a, b, tw, mult = term
            av = C[a] * mult
            assert av < (1 << 384)  # the scaled operand fits twelve limbs
            bv = T5[b] if tw else C[b]
            acc = av * ((bv >> (96 * j)) & ((1 << 96) - 1))
        assert acc < (1 << (32 * 16))
        P[L] = acc
    F = 0
    T = 0
    for j in range(4):
        Qj = sum(P[4 * j + g] for g in range(4))
        assert Qj < (1 << (32 * 16))
        T += Qj << (96 * j)
        ql = limbs(Qj, 16)
        for g in range(4):
            f = sum(ql[g + 4 * t] * D[3 * j + g + 4 * t] for t in range(4))
            assert f < (1 << (32 * 13))
            F += f
    assert F < (1 << 416)
    c = barrett(F)
    assert c % Q == T * pow(RM, -1, Q) % Q
    return c


def check_csqr(u_mont, s_mont):
    """u_mont: Montgomery coefficients of a cyclotomic-subgroup element times the scale, any representatives in [0, 2q];
    s_mont: the scale (Montgomery).  The round's output is the square, scale s^2."""
    table = csqr_table()
    C = list(u_mont) + [s_mont]
    out = [csqr_row(C, k, table) for k in range(13)]
    rinv = pow(RM, -1, Q)
    want = [x * rinv % Q for x in fq12_mul_ref([c % Q for c in u_mont], [c % Q for c in u_mont])]
    assert [c % Q for c in out[:12]] == want
    assert out[12] % Q == s_mont * s_mont * rinv % Q
    for c in out[:12]:
        twin5(c)
    return out


def cyclotomic_element(rnd):
    """a random element of the cyclotomic subgroup: f^((q^6 - 1)(q^2 + 1)), canonical coefficients"""
    def mul(a, b):
        return fq12_mul_ref(a, b)

    def power(a, e):
        r = [1] + [0] * 11
        for bit in bin(e)[2:]:
            r = mul(r, r)
            if bit == "1":
                r = mul(r, a)
        return r
    f = [rnd.randrange(Q) for _ in range(12)]
    return power(f, (Q**6 - 1) * (Q**2 + 1))


def self_test_csqr(n_random=3, seed=2):
    rnd = random.Random(seed)
    for it in range(n_random):
        u = cyclotomic_element(rnd)
        s = rnd.randrange(1, Q) if it else 1
        um = [(x * s % Q) * RM % Q for x in u]  # scaled, Montgomery
        sm = s * RM % Q
        out = check_csqr(um, sm)
        # other representatives of the same residues (a round leaves values in [0, 2q)), and a second squaring
        um2 = [x + Q if (x + Q <= 2 * Q and rnd.random() < 0.5) else x for x in um]
        check_csqr(um2, sm + Q if rnd.random() < 0.5 else sm)
        check_csqr(out[:12], out[12])
    # bounds at the operands' upper limits (every assertion inside csqr_row is a bound or the row's own congruence)
    table = csqr_table()
    for k in range(13):
        csqr_row([2 * Q] * 13, k, table)
    return True


def self_test(n_random=40, seed=1):
    rnd = random.Random(seed)
    stats = {}
    assert all(d < Q for d in D) and MU < (1 << 40)
    # worst case: every operand at its upper bound
    check([2 * Q] * 12, [2 * Q] * 12, stats)
    check([2 * Q] * 12, [0] * 12, stats)
    check([0] * 12, [2 * Q] * 12, stats)
    check([RM - 1 if False else 2 * Q - 1] * 12, [1] * 12, stats)
    for _ in range(n_random):
        A = [rnd.randrange(0, 2 * Q + 1) for _ in range(12)]
        B = [rnd.randrange(0, 2 * Q + 1) for _ in range(12)]
        C = check(A, B, stats)
        check(C, C, stats)  # a round's outputs are valid inputs
    return stats


if __name__ == "__main__":
    print(self_test())
    print("cyclotomic squaring round:", self_test_csqr())
