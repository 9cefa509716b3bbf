#!/usr/bin/env python3
"""Derive every BLS12-377 constant from the seed x and emit C headers.

Writes
  oracle/constants.h                 (CPU oracle, test infrastructure)
  snarkos_amd/csrc/constants.hpp     (product: HIP + host code)
Both hold the same numbers as little-endian 64-bit limbs; field elements are in
Montgomery form (R = 2^256 for Fr, 2^384 for Fq) exactly as snarkVM keeps them in
memory.  Nothing here is copied from anywhere:
the values are recomputed on every run and self-checked.
"""
import os
import sys

X = 0x8508C00000000001
R = X**4 - X**2 + 1
Q = (X - 1) ** 2 * R // 3 + X

G1X = 81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695
G1Y = 241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030
G2X = (233578398248691099356572568220835526895379068987715365179118596935057653620464273615301663571204657964920925606294,
       140913150380207355837477652521042157274541796891053068589147167627541651775299824604154852141315666357241556069118)
G2Y = (63160294768292073209381361943935198908131692476676907196754037919244929611450776219210369229519898517858833747423,
       149157405641012693445398062341192467754805999074082136895788947234480009303640899064710353187729182149407503257491)


def is_probable_prime(n):
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % a == 0:
            return n == a
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


assert is_probable_prime(R) and is_probable_prime(Q)
assert (G1Y * G1Y - G1X**3 - 1) % Q == 0
assert (Q - 1) % 12 == 0
assert pow(-5 % Q, (Q - 1) // 2, Q) == Q - 1  # -5 is a non-residue
assert pow(22, (R - 1) // 2, R) == R - 1  # 22 is a non-residue in Fr
assert (R - 1) % (1 << 47) == 0 and (R - 1) % (1 << 48) != 0

FR_R = (1 << 256) % R
FQ_R = (1 << 384) % Q
FR_INV = (-pow(R, -1, 1 << 64)) % (1 << 64)
FQ_INV = (-pow(Q, -1, 1 << 64)) % (1 << 64)
assert FR_INV == 0x0A117FFFFFFFFFFF and FQ_INV == 0x8508BFFFFFFFFFFF
FR_ROOT = pow(22, (R - 1) >> 47, R)
# Frobenius on Fq12 = Fq[w]/(w^12+5):  (w^k)^(q^e) = gamma_e^k * w^k with gamma_e in Fq
GAMMA = [pow(-5 % Q, (Q**e - 1) // 12, Q) for e in (1, 2, 3)]
# twist coefficient b' = 1/u = -u/5  ->  (0, -1/5)
TWIST_B = (0, (-pow(5, -1, Q)) % Q)
assert (TWIST_B[1] * (-5) % Q) == 1  # (b1 u) * u = -5 b1 = 1


# GLV endomorphism on G1 (y^2 = x^3 + 1): phi(x, y) = (beta x, y) acts as multiplication by lambda = x^2 - 1 on the
# order-r subgroup (lambda^2 + lambda + 1 = r); beta is the cube root of unity in Fq that matches this lambda, found
# by comparing lambda G with (beta Gx, Gy).  GLV_MU = floor(2^256 / lambda) for the Barrett quotient s div lambda.
LAMBDA = X * X - 1
assert (LAMBDA * LAMBDA + LAMBDA + 1) == R


def _ec_add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    (x1, y1), (x2, y2) = p1, p2
    if x1 == x2:
        if (y1 + y2) % Q == 0:
            return None
        l = 3 * x1 * x1 * pow(2 * y1, -1, Q) % Q
    else:
        l = (y2 - y1) * pow(x2 - x1, -1, Q) % Q
    x3 = (l * l - x1 - x2) % Q
    return (x3, (l * (x1 - x3) - y1) % Q)


def _ec_mul(k, p):
    acc = None
    while k:
        if k & 1:
            acc = _ec_add(acc, p)
        p = _ec_add(p, p)
        k >>= 1
    return acc


def _glv_beta():
    g = 2
    while pow(g, (Q - 1) // 3, Q) == 1:
        g += 1
    b = pow(g, (Q - 1) // 3, Q)
    lam_g = _ec_mul(LAMBDA, (G1X, G1Y))
    for cand in (b, b * b % Q):
        if lam_g == (cand * G1X % Q, G1Y):
            return cand
    raise AssertionError("no cube root of unity matches lambda")


BETA = _glv_beta()
assert pow(BETA, 3, Q) == 1 and BETA != 1
GLV_MU = (1 << 256) // LAMBDA
assert LAMBDA.bit_length() == 127 and GLV_MU.bit_length() <= 160


# Poseidon over Fq: width 3, x^17, 8 full + 31 partial rounds; constants generated here
# (SHA-256 counter stream, Cauchy MDS) because snarkVM's are not available offline.
import hashlib  # noqa: E402

POSEIDON_TAG = b"MI355X-Poseidon-BLS12-377-Fq-v1"
POSEIDON_ROUNDS = 39
POSEIDON_ARK = [[int.from_bytes(hashlib.sha256(POSEIDON_TAG + bytes([r, i]) + b"\x00").digest()
                                + hashlib.sha256(POSEIDON_TAG + bytes([r, i]) + b"\x01").digest(), "little") % Q
                 for i in range(3)] for r in range(POSEIDON_ROUNDS)]
POSEIDON_MDS = [[pow(i + 3 + j, -1, Q) for j in range(3)] for i in range(3)]


FQ_T = (Q - 1) >> 46
assert (Q - 1) == FQ_T << 46 and FQ_T & 1 and pow(15, (Q - 1) // 2, Q) == Q - 1


def limbs(v, n):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def arr(v, n):
    return "{" + ", ".join("0x%016xULL" % l for l in limbs(v, n)) + "}"


def fq(v):
    return arr(v * FQ_R % Q, 6)


def fr(v):
    return arr(v * FR_R % R, 4)


def emit(prefix, static_kw):
    o = []
    a = o.append
    a("/* GENERATED by tools/gen_constants.py -- do not edit.  BLS12-377, derived from x = 0x%x. */" % X)
    a("#pragma once")
    a("#include <stdint.h>")
    a("#define %sBLS_X 0x%016xULL" % (prefix, X))
    a("#define %sFR_INV 0x%016xULL" % (prefix, FR_INV))
    a("#define %sFQ_INV 0x%016xULL" % (prefix, FQ_INV))
    a("#define %sFR_TWO_ADICITY 47" % prefix)
    a("%s uint64_t %sFR_MODULUS[4] = %s;" % (static_kw, prefix, arr(R, 4)))
    a("%s uint64_t %sFR_ONE[4] = %s;   /* R mod r */" % (static_kw, prefix, arr(FR_R, 4)))
    a("%s uint64_t %sFR_R2[4] = %s;" % (static_kw, prefix, arr(FR_R * FR_R % R, 4)))
    a("%s uint64_t %sFR_ROOT_OF_UNITY[4] = %s; /* order 2^47, Montgomery */" % (static_kw, prefix, fr(FR_ROOT)))
    a("%s uint64_t %sFR_GENERATOR[4] = %s; /* 22, Montgomery */" % (static_kw, prefix, fr(22)))
    a("%s uint64_t %sFR_GENERATOR_INV[4] = %s;" % (static_kw, prefix, fr(pow(22, -1, R))))
    a("%s uint64_t %sFQ_MODULUS[6] = %s;" % (static_kw, prefix, arr(Q, 6)))
    a("%s uint64_t %sFQ_ONE[6] = %s;   /* R mod q */" % (static_kw, prefix, arr(FQ_R, 6)))
    a("%s uint64_t %sFQ_R2[6] = %s;" % (static_kw, prefix, arr(FQ_R * FQ_R % Q, 6)))
    a("%s uint64_t %sG1_GEN_X[6] = %s;" % (static_kw, prefix, fq(G1X)))
    a("%s uint64_t %sG1_GEN_Y[6] = %s;" % (static_kw, prefix, fq(G1Y)))
    a("%s uint64_t %sG2_GEN_X[2][6] = {%s, %s};" % (static_kw, prefix, fq(G2X[0]), fq(G2X[1])))
    a("%s uint64_t %sG2_GEN_Y[2][6] = {%s, %s};" % (static_kw, prefix, fq(G2Y[0]), fq(G2Y[1])))
    a("%s uint64_t %sTWIST_B[2][6] = {%s, %s};" % (static_kw, prefix, fq(TWIST_B[0]), fq(TWIST_B[1])))
    a("/* GLV: phi(x, y) = (beta x, y) = lambda (x, y) on G1; lambda = x^2 - 1 (127 bits); MU = floor(2^256 / lambda) */")
    a("%s uint64_t %sGLV_BETA[6] = %s; /* Montgomery */" % (static_kw, prefix, fq(BETA)))
    a("%s uint64_t %sGLV_LAMBDA[2] = %s;" % (static_kw, prefix, arr(LAMBDA, 2)))
    a("%s uint64_t %sGLV_MU[3] = %s;" % (static_kw, prefix, arr(GLV_MU, 3)))
    a("/* square roots in Fq (Tonelli-Shanks): q - 1 = 2^46 t, t odd; FQ_T_MINUS1_HALF = (t - 1) / 2 (330 bits);")
    a(" * FQ_2SYLOW_GEN = 15^t (15 is a quadratic non-residue): a generator of the 2^46-th roots of unity, Montgomery;")
    a(" * FQ_HALF = (q - 1) / 2: y is the 'larger' root of its pair when its canonical value exceeds this */")
    a("#define %sFQ_TWO_ADICITY 46" % prefix)
    a("%s uint64_t %sFQ_T_MINUS1_HALF[6] = %s;" % (static_kw, prefix, arr((FQ_T - 1) // 2, 6)))
    a("%s uint64_t %sFQ_2SYLOW_GEN[6] = %s;" % (static_kw, prefix, fq(pow(15, FQ_T, Q))))
    a("%s uint64_t %sFQ_HALF[6] = %s;" % (static_kw, prefix, arr((Q - 1) // 2, 6)))
    a("/* FQ_2SYLOW_POW[j] = FQ_2SYLOW_GEN^(2^j), j < 46 (Montgomery): every g and c of the Tonelli-Shanks loop is one of these */")
    a("%s uint64_t %sFQ_2SYLOW_POW[46][6] = {" % (static_kw, prefix))
    for j in range(46):
        a("  %s," % fq(pow(15, FQ_T << j, Q)))
    a("};")
    a("%s uint64_t %sFR_MODULUS_BITS[4] = %s; /* r itself: the subgroup check multiplies by it */" % (static_kw, prefix, arr(R, 4)))
    a("/* FROB[e-1][k] = gamma_e^k (Montgomery): multiply the Fq coefficient of w^k, k = i + 2j + 6b for c_i.c_j.c_b */")
    a("%s uint64_t %sFROB[3][12][6] = {" % (static_kw, prefix))
    for e in range(3):
        a("  {" + ",\n   ".join(fq(pow(GAMMA[e], k, Q)) for k in range(12)) + "},")
    a("};")
    a("/* Poseidon over Fq: width 3 (rate 2, capacity 1), S-box x^17, 4 + 31 + 4 rounds; Montgomery */")
    a("#define %sPOSEIDON_ROUNDS 39" % prefix)
    a("#define %sPOSEIDON_FULL_HALF 4" % prefix)
    a("#define %sPOSEIDON_PARTIAL 31" % prefix)
    a("%s uint64_t %sPOSEIDON_ARK[39][3][6] = {" % (static_kw, prefix))
    for r in range(POSEIDON_ROUNDS):
        a("  {" + ", ".join(fq(POSEIDON_ARK[r][i]) for i in range(3)) + "},")
    a("};")
    a("%s uint64_t %sPOSEIDON_MDS[3][3][6] = {" % (static_kw, prefix))
    for i in range(3):
        a("  {" + ", ".join(fq(POSEIDON_MDS[i][j]) for j in range(3)) + "},")
    a("};")
    return "\n".join(o) + "\n"


# ---- Edwards-BLS12 (Schnorr signatures of batch certificates): twisted Edwards a = -1, d = 3021 over
# Fr [RECALL for the parameters; checked below], subgroup order s, cofactor 4.  The generator and the Poseidon constants
# over Fr are this repo's own (snarkVM's are not available offline): same derivation as oracle/pyref.py, written again.
ED_D = 3021
ED_S = 2111115437357092606062206234695386632838870926408408195193685246394721360383
assert is_probable_prime(ED_S) and abs(4 * ED_S - (R + 1)) <= 2 * int(R ** 0.5) + 2
assert pow(R - 1, (R - 1) // 2, R) == 1 and pow(ED_D, (R - 1) // 2, R) == R - 1  # a square, d not: complete addition law


def _ed_add(p, q):
    (x1, y1), (x2, y2) = p, q
    t = ED_D * x1 * x2 * y1 * y2 % R
    return ((x1 * y2 + y1 * x2) * pow(1 + t, -1, R) % R, (y1 * y2 + x1 * x2) * pow(1 - t, -1, R) % R)


def _ed_mul(p, k):
    acc = (0, 1)
    while k:
        if k & 1:
            acc = _ed_add(acc, p)
        p = _ed_add(p, p)
        k >>= 1
    return acc


def _fr_sqrt(v):
    if pow(v, (R - 1) // 2, R) != 1:
        return None
    t = (R - 1) >> 47
    m, c, x, b = 47, pow(22, t, R), pow(v, (t + 1) // 2, R), pow(v, t, R)
    while b != 1:
        i, b2 = 0, b
        while b2 != 1:
            b2, i = b2 * b2 % R, i + 1
        g = pow(c, 1 << (m - i - 1), R)
        m, c, x, b = i, g * g % R, x * g % R, b * g * g % R
    return x


def _ed_generator():
    x = 2
    while True:
        y = _fr_sqrt((1 + x * x) * pow(1 - ED_D * x * x, -1, R) % R)
        if y is not None:
            g = _ed_mul((x, min(y, R - y)), 4)
            if g != (0, 1):
                assert _ed_mul(g, ED_S) == (0, 1)
                return g
        x += 1


ED_GEN = _ed_generator()
ED_TAG = b"MI355X-Poseidon-BLS12-377-Fr-v1"


def _ed_ark(width):
    return [[int.from_bytes(hashlib.sha256(ED_TAG + bytes([width, r, i]) + b"\x00").digest()
                            + hashlib.sha256(ED_TAG + bytes([width, r, i]) + b"\x01").digest(), "little") % R
             for i in range(width)] for r in range(POSEIDON_ROUNDS)]


def emit_ed():
    o = []
    a = o.append
    a("/* GENERATED by tools/gen_constants.py -- do not edit.  Edwards-BLS12 over Fr of BLS12-377: -x^2 + y^2 = 1 + d x^2 y^2. */")
    a("#pragma once")
    a("#include <stdint.h>")
    a("static const uint64_t MI_ED_D[4] = %s; /* 3021, Montgomery */" % fr(ED_D))
    a("static const uint64_t MI_ED_2D[4] = %s;" % fr(2 * ED_D))
    a("static const uint64_t MI_ED_ORDER[4] = %s; /* the subgroup order s (251 bits), canonical */" % arr(ED_S, 4))
    a("static const uint64_t MI_ED_GEN[2][4] = {%s, %s}; /* own generator: 4 (x, y), smallest x >= 2 on the curve */" % (fr(ED_GEN[0]), fr(ED_GEN[1])))
    for w in (5, 9):
        ark = _ed_ark(w)
        a("/* Poseidon over Fr, width %d (rate %d): x^17, 4 + 31 + 4 rounds, own constants (Montgomery) */" % (w, w - 1))
        a("static const uint64_t MI_ED_ARK%d[39][%d][4] = {" % (w, w))
        for r in range(POSEIDON_ROUNDS):
            a("  {" + ", ".join(fr(ark[r][i]) for i in range(w)) + "},")
        a("};")
        a("static const uint64_t MI_ED_MDS%d[%d][%d][4] = {" % (w, w, w))
        for i in range(w):
            a("  {" + ", ".join(fr(pow(i + w + j, -1, R)) for j in range(w)) + "},")
        a("};")
    return "\n".join(o) + "\n"


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "snarkos_amd", "csrc", "ed_constants.hpp"), "w") as f:
        f.write(emit_ed())
    with open(os.path.join(root, "oracle", "constants.h"), "w") as f:
        f.write(emit("ORC_", "static const"))
    os.makedirs(os.path.join(root, "snarkos_amd", "csrc"), exist_ok=True)
    with open(os.path.join(root, "snarkos_amd", "csrc", "constants.hpp"), "w") as f:
        f.write(emit("MI_", "static const"))
    print("constants written")


if __name__ == "__main__":
    sys.exit(main())
