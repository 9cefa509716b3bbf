"""Independent Python big-int restatement of the BLS12-377 arithmetic on the hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under snarkos_amd/ may import this file; it is
used by tests/ and by oracle/gen_golden.py to pin the C oracle (oracle/*.c).

PARITY UNPINNED: the reference's implementation of this path lives in the
un-vendored crate family snarkVM 1.0.0 @ git rev dea322b (Cargo.toml:47-52,
Cargo.lock:3578-3580 of the reference); no golden vector for MSM / NTT / KZG /
pairing exists in the reference tree.  The oracle is
therefore pinned by mathematics: every constant below is *derived* from the BLS12
seed x, and every algorithm is the textbook definition (affine chord-and-tangent,
O(n^2) DFT, generic Miller loop over E(Fq12) in the power basis Fq[w]/(w^12+5),
final exponentiation by one big pow()).  The C oracle uses different algorithms
(Montgomery limbs, Jacobian coordinates, tower Fq2/Fq6/Fq12, sparse lines,
x-chains), so agreement between the two is a real cross-check.

Conventions that follow the snarkVM surface named in SURVEY.md Appendix B
(VariableBase::msm, EvaluationDomain::{fft,ifft}, KZG10::{check,batch_check}):
  * MSM: sum_i scalars[i] * bases[i], scalars canonical integers < r.
  * NTT: natural order in/out, omega_n = ROOT^(2^(47-log n)), ROOT = 22^((r-1)/2^47);
    inverse scales by n^-1; coset shift by the multiplicative generator 22.
  * KZG10 single check:  e(C - v*G, H) == e(W, tauH - z*H)
    batch check with randomizers r_i:
       e(sum r_i (C_i - v_i G + z_i W_i), H) * e(-sum r_i W_i, tauH) == 1
"""
import hashlib

# ----------------------------------------------------------------------------- constants
X = 0x8508C00000000001
R = X**4 - X**2 + 1
Q = (X - 1) ** 2 * R // 3 + X
assert R == 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001
assert Q == 0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001

G1_GEN = (
    81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695,
    241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030,
)
G2_GEN = (
    (233578398248691099356572568220835526895379068987715365179118596935057653620464273615301663571204657964920925606294,
     140913150380207355837477652521042157274541796891053068589147167627541651775299824604154852141315666357241556069118),
    (63160294768292073209381361943935198908131692476676907196754037919244929611450776219210369229519898517858833747423,
     149157405641012693445398062341192467754805999074082136895788947234480009303640899064710353187729182149407503257491),
)
FR_TWO_ADICITY = 47
FR_GENERATOR = 22
FR_ROOT = pow(FR_GENERATOR, (R - 1) >> FR_TWO_ADICITY, R)
assert FR_ROOT == 8065159656716812877374967518403273466521432693661810619979959746626482506078
NONRES = -5  # Fq2 = Fq[u]/(u^2+5); Fq12 = Fq[w]/(w^12+5) with u = w^6, v = w^2

FR_MONT_R = (1 << 256) % R
FQ_MONT_R = (1 << 384) % Q


def inv(a, p):
    return pow(a, p - 2, p)


# ----------------------------------------------------------------------------- Fq2
def fq2_add(a, b):
    return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)


def fq2_sub(a, b):
    return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)


def fq2_mul(a, b):
    return ((a[0] * b[0] + NONRES * a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)


def fq2_inv(a):
    n = inv((a[0] * a[0] - NONRES * a[1] * a[1]) % Q, Q)
    return (a[0] * n % Q, (-a[1]) * n % Q)


def fq2_neg(a):
    return ((-a[0]) % Q, (-a[1]) % Q)


# ----------------------------------------------------------------------------- Fq12, power basis
def f12_one():
    return [1] + [0] * 11


def f12_mul(a, b):
    t = [0] * 23
    for i in range(12):
        ai = a[i]
        if ai:
            for j in range(12):
                t[i + j] += ai * b[j]
    return [(t[k] + NONRES * t[k + 12]) % Q if k < 11 else t[k] % Q for k in range(12)]


def f12_pow(a, e):
    r = f12_one()
    for bit in bin(e)[2:]:
        r = f12_mul(r, r)
        if bit == "1":
            r = f12_mul(r, a)
    return r


def f12_from_fq2(c, shift):
    """c0 + c1*u placed at w^shift: u = w^6."""
    out = [0] * 12
    out[shift % 12] = c[0]
    k = shift + 6
    if k >= 12:
        out[k - 12] = (out[k - 12] + NONRES * c[1]) % Q
    else:
        out[k] = c[1]
    return out


def f12_add(a, b):
    return [(x + y) % Q for x, y in zip(a, b)]


def f12_sub(a, b):
    return [(x - y) % Q for x, y in zip(a, b)]


def f12_inv(a):
    # a^(q^12-2) is far too slow; use linear algebra-free trick: norm down via conjugates is
    # overkill here.  Solve a * b = 1 by Gaussian elimination over Fq (12x12).
    n = 12
    M = [[0] * (n + 1) for _ in range(n)]
    for j in range(n):
        e = [0] * 12
        e[j] = 1
        col = f12_mul(a, e)
        for i in range(n):
            M[i][j] = col[i]
    M[0][n] = 1
    for c in range(n):
        p = next(i for i in range(c, n) if M[i][c] % Q)
        M[c], M[p] = M[p], M[c]
        iv = inv(M[c][c], Q)
        M[c] = [x * iv % Q for x in M[c]]
        for i in range(n):
            if i != c and M[i][c]:
                f = M[i][c]
                M[i] = [(x - f * y) % Q for x, y in zip(M[i], M[c])]
    return [M[i][n] for i in range(n)]


# ----------------------------------------------------------------------------- G1 (affine, None = infinity)
def g1_is_on_curve(P):
    if P is None:
        return True
    x, y = P
    return (y * y - x * x * x - 1) % Q == 0


def g1_neg(P):
    return None if P is None else (P[0], (-P[1]) % Q)


def g1_add(P, S):
    if P is None:
        return S
    if S is None:
        return P
    x1, y1 = P
    x2, y2 = S
    if x1 == x2:
        if (y1 + y2) % Q == 0:
            return None
        lam = 3 * x1 * x1 * inv(2 * y1, Q) % Q
    else:
        lam = (y2 - y1) * inv((x2 - x1) % Q, Q) % Q
    x3 = (lam * lam - x1 - x2) % Q
    return (x3, (lam * (x1 - x3) - y1) % Q)


def g1_mul(P, k):
    k %= R
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = g1_add(acc, acc)
        if bit == "1":
            acc = g1_add(acc, P)
    return acc


# ----------------------------------------------------------------------------- compressed G1 points
# snarkVM's CanonicalSerialize of a short-Weierstrass affine point, compressed [RECALL: snarkvm-curves
# templates/short_weierstrass_jacobian + SWFlags; not in /root/reference]: 48 bytes, x as a little-endian integer below q
# (377 bits); bit 7 of the last byte is set when y is the LARGER of the two roots (y > (q - 1) / 2 as integers),
# bit 6 marks the point at infinity (x = 0 then).  Big-integer restatement, independent of the C oracle.
def fq_sqrt(a):
    """a square root of a mod Q (Tonelli-Shanks over the 2^46-th roots of unity), or None"""
    a %= Q
    if a == 0:
        return 0
    if pow(a, (Q - 1) // 2, Q) != 1:
        return None
    s, t = 0, Q - 1
    while t % 2 == 0:
        s, t = s + 1, t // 2
    z = pow(15, t, Q)
    m, c, x, b = s, z, pow(a, (t + 1) // 2, Q), pow(a, t, Q)
    while b != 1:
        i, b2 = 0, b
        while b2 != 1:
            b2, i = b2 * b2 % Q, i + 1
        g = pow(c, 1 << (m - i - 1), Q)
        m, c, x, b = i, g * g % Q, x * g % Q, b * g * g % Q
    return x


def g1_compress(P):
    if P is None:
        return bytes(47) + bytes([0x40])
    x, y = P
    out = bytearray(x.to_bytes(48, "little"))
    if y > (Q - 1) // 2:
        out[47] |= 0x80
    return bytes(out)


def g1_decompress(data, check_subgroup=True):
    """(ok, point): ok False for x >= q, a non-residue x^3 + 1, a malformed infinity, or (checked) a point outside the
    order-r subgroup"""
    assert len(data) == 48
    flags = data[47] & 0xC0
    x = int.from_bytes(bytes(data[:47]) + bytes([data[47] & 0x3F]), "little")
    if flags & 0x40:
        return (x == 0 and not flags & 0x80), None
    if x >= Q:
        return False, None
    y = fq_sqrt(x * x * x + 1)
    if y is None:
        return False, None
    if y == 0 and flags & 0x80:
        return False, None  # y = 0 is its own negative: the "larger root" flag has nothing to select (non-canonical)
    if (y > (Q - 1) // 2) != bool(flags & 0x80):
        y = (Q - y) % Q
    P = (x, y)
    if check_subgroup and _g1_mul_any(P, R) is not None:
        return False, None
    return True, P


def _g1_mul_any(P, k):
    """k P without reducing k mod r (P need not lie in the order-r subgroup)"""
    acc = None
    for bit in bin(k)[2:]:
        acc = g1_add(acc, acc)
        if bit == "1":
            acc = g1_add(acc, P)
    return acc


def msm(bases, scalars):
    """VariableBase::msm semantic: sum scalars[i]*bases[i] over min(len) items."""
    acc = None
    for P, s in zip(bases, scalars):
        acc = g1_add(acc, g1_mul(P, s))
    return acc


# ----------------------------------------------------------------------------- G2 on the D-twist y^2 = x^3 + 1/u
TWIST_B = fq2_inv((0, 1))


def g2_is_on_curve(P):
    if P is None:
        return True
    x, y = P
    return fq2_sub(fq2_mul(y, y), fq2_add(fq2_mul(fq2_mul(x, x), x), TWIST_B)) == (0, 0)


def g2_add(P, S):
    if P is None:
        return S
    if S is None:
        return P
    x1, y1 = P
    x2, y2 = S
    if x1 == x2:
        if fq2_add(y1, y2) == (0, 0):
            return None
        xx = fq2_mul(x1, x1)
        lam = fq2_mul(fq2_add(fq2_add(xx, xx), xx), fq2_inv(fq2_add(y1, y1)))
    else:
        lam = fq2_mul(fq2_sub(y2, y1), fq2_inv(fq2_sub(x2, x1)))
    x3 = fq2_sub(fq2_sub(fq2_mul(lam, lam), x1), x2)
    return (x3, fq2_sub(fq2_mul(lam, fq2_sub(x1, x3)), y1))


def g2_mul(P, k):
    k %= R
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = g2_add(acc, acc)
        if bit == "1":
            acc = g2_add(acc, P)
    return acc


# ----------------------------------------------------------------------------- pairing (generic, E(Fq12))
def _untwist(Qp):
    """psi: E'(Fq2) -> E(Fq12), (x', y') -> (x' w^2, y' w^3)."""
    return (f12_from_fq2(Qp[0], 2), f12_from_fq2(Qp[1], 3))


def miller_loop(P, Qp):
    """f_{x,psi(Q)}(P): ate Miller loop over the seed x with affine lines in Fq12."""
    if P is None or Qp is None:
        return f12_one()
    xq, yq = _untwist(Qp)
    xp = [P[0]] + [0] * 11
    yp = [P[1]] + [0] * 11
    T = (xq, yq)
    f = f12_one()

    def line(T1, T2):
        x1, y1 = T1
        x2, y2 = T2
        if x1 == x2 and y1 == y2:
            xx = f12_mul(x1, x1)
            lam = f12_mul(f12_add(f12_add(xx, xx), xx), f12_inv(f12_add(y1, y1)))
        else:
            lam = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
        x3 = f12_sub(f12_sub(f12_mul(lam, lam), x1), x2)
        y3 = f12_sub(f12_mul(lam, f12_sub(x1, x3)), y1)
        # l(P) = yP - y1 - lam (xP - x1)
        val = f12_sub(f12_sub(yp, y1), f12_mul(lam, f12_sub(xp, x1)))
        return val, (x3, y3)

    for bit in bin(X)[3:]:
        l, T = line(T, T)
        f = f12_mul(f12_mul(f, f), l)
        if bit == "1":
            l, T = line(T, (xq, yq))
            f = f12_mul(f, l)
    return f


FINAL_EXP = (Q**12 - 1) // R
# The x-chain used by optimised implementations computes the cube of the reduced pairing:
#   3*(q^4-q^2+1)/r = (x-1)^2 (x+q) (x^2+q^2-1) + 3
assert 3 * ((Q**4 - Q**2 + 1) // R) == (X - 1) ** 2 * (X + Q) * (X * X + Q * Q - 1) + 3


def final_exponentiation(f, cubed=True):
    return f12_pow(f, FINAL_EXP * (3 if cubed else 1))


def pairing(P, Qp, cubed=True):
    return final_exponentiation(miller_loop(P, Qp), cubed)


def pairing_product_is_one(pairs):
    f = f12_one()
    for P, Qp in pairs:
        f = f12_mul(f, miller_loop(P, Qp))
    return final_exponentiation(f, cubed=False) == f12_one()


# ----------------------------------------------------------------------------- NTT over Fr
def omega(log_n):
    return pow(FR_ROOT, 1 << (FR_TWO_ADICITY - log_n), R)


def dft(a, w):
    n = len(a)
    return [sum(a[j] * pow(w, i * j, R) for j in range(n)) % R for i in range(n)]


def ntt(a, inverse=False, coset=False):
    """EvaluationDomain::{fft, ifft, coset_fft, coset_ifft} semantics, natural order."""
    n = len(a)
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    a = list(a)
    w = omega(log_n)
    if not inverse:
        if coset:
            g = 1
            for i in range(n):
                a[i] = a[i] * g % R
                g = g * FR_GENERATOR % R
        return _fft(a, w)
    out = _fft(a, inv(w, R))
    ninv = inv(n, R)
    out = [x * ninv % R for x in out]
    if coset:
        gi = inv(FR_GENERATOR, R)
        g = 1
        for i in range(n):
            out[i] = out[i] * g % R
            g = g * gi % R
    return out


def _fft(a, w):
    n = len(a)
    if n == 1:
        return a
    ev = _fft(a[0::2], w * w % R)
    od = _fft(a[1::2], w * w % R)
    out = [0] * n
    t = 1
    for i in range(n // 2):
        o = od[i] * t % R
        out[i] = (ev[i] + o) % R
        out[i + n // 2] = (ev[i] - o) % R
        t = t * w % R
    return out


# ----------------------------------------------------------------------------- KZG10
def kzg_check(vk_tau_h, C, z, v, W):
    """e(C - v G, H) == e(W, tauH - zH), written as a pairing product."""
    lhs = g1_add(C, g1_neg(g1_mul(G1_GEN, v)))
    lhs = g1_add(lhs, g1_mul(W, z))
    return pairing_product_is_one([(lhs, G2_GEN), (g1_neg(W), vk_tau_h)])


def kzg_batch_check(vk_tau_h, Cs, zs, vs, Ws, rs):
    total_c = None
    total_w = None
    gs = 0
    for C, z, v, W, r in zip(Cs, zs, vs, Ws, rs):
        total_c = g1_add(total_c, g1_mul(g1_add(C, g1_mul(W, z)), r))
        gs = (gs + r * v) % R
        total_w = g1_add(total_w, g1_mul(W, r))
    total_c = g1_add(total_c, g1_neg(g1_mul(G1_GEN, gs)))
    return pairing_product_is_one([(total_c, G2_GEN), (g1_neg(total_w), vk_tau_h)])


def kzg_batch_check_hiding(vk_tau_h, gamma_g, Cs, zs, vs, Ws, rbars, rs):
    """KZG10::batch_check with hiding: commitments C = p(tau) G + rho(tau) gamma_g, proofs (w, random_v = rho(z))"""
    total_c = None
    total_w = None
    gs = hs = 0
    for C, z, v, W, rb, r in zip(Cs, zs, vs, Ws, rbars, rs):
        total_c = g1_add(total_c, g1_mul(g1_add(C, g1_mul(W, z)), r))
        gs = (gs + r * v) % R
        hs = (hs + r * rb) % R
        total_w = g1_add(total_w, g1_mul(W, r))
    total_c = g1_add(total_c, g1_neg(g1_mul(G1_GEN, gs)))
    total_c = g1_add(total_c, g1_neg(g1_mul(gamma_g, hs)))
    return pairing_product_is_one([(total_c, G2_GEN), (g1_neg(total_w), vk_tau_h)])


# ----------------------------------------------------------------------------- byte / limb helpers (ABI layout)
def to_limbs(v, n):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def from_limbs(l):
    return sum(int(x) << (64 * i) for i, x in enumerate(l))


def fq_to_mont(a):
    return a * FQ_MONT_R % Q


def fq_from_mont(a):
    return a * inv(FQ_MONT_R, Q) % Q


def fr_to_mont(a):
    return a * FR_MONT_R % R


def fr_from_mont(a):
    return a * inv(FR_MONT_R, R) % R


def sha256(b):
    return hashlib.sha256(b).digest()


# ----------------------------------------------------------------------------- splitmix64 (shared PRNG spec)
class SplitMix64:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def below(self, p, limbs):
        """Uniform-ish value < p: draw `limbs` 64-bit words (little-endian), reduce mod p."""
        v = 0
        for i in range(limbs):
            v |= self.next() << (64 * i)
        return v % p


# ----------------------------------------------------------------------------- VarunaShaped-v1
VS_TAG = b"MI355X-VarunaShaped-v1".ljust(32, b"\0")
VS_INPUTS, VS_PROOF_COMMS, VS_VK_COMMS, VS_EVALS = 16, 9, 12, 21
VS_PROOF_BYTES, VS_CIRCUIT_VK_BYTES = 2256, 1184


def _rd_fr(b):
    return fr_from_mont(int.from_bytes(b, "little"))


def _rd_g1(b):
    if not any(b):
        return None
    return (fq_from_mont(int.from_bytes(b[:48], "little")), fq_from_mont(int.from_bytes(b[48:96], "little")))


def vs_parse_proof(b):
    assert len(b) == VS_PROOF_BYTES
    o = 16
    d = {"circuit": int.from_bytes(b[:4], "little"), "raw": bytes(b)}
    d["inputs"] = [_rd_fr(b[o + 32 * i:o + 32 * i + 32]) for i in range(VS_INPUTS)]
    o += 32 * VS_INPUTS
    d["comms"] = [_rd_g1(b[o + 96 * i:o + 96 * i + 96]) for i in range(VS_PROOF_COMMS)]
    o += 96 * VS_PROOF_COMMS
    d["evals"] = [_rd_fr(b[o + 32 * i:o + 32 * i + 32]) for i in range(VS_EVALS)]
    o += 32 * VS_EVALS
    d["w"] = [_rd_g1(b[o + 96 * i:o + 96 * i + 96]) for i in range(2)]
    return d


def vs_parse_circuit(b):
    assert len(b) == VS_CIRCUIT_VK_BYTES
    return {"comms": [_rd_g1(b[96 * i:96 * i + 96]) for i in range(VS_VK_COMMS)], "digest": bytes(b[1152:1184])}


# Fiat-Shamir transcript of the synthetic shape: SHA-256 over the in-memory bytes (round 1), or a Poseidon sponge over
# Fq with the reference's absorb volume (37 permutations per proof).  A process-wide switch, like
# the C oracle's: test infrastructure, never used concurrently with two kinds.
VS_FS_SHA256, VS_FS_POSEIDON = 0, 1
_VS_FS = [VS_FS_SHA256]


def vs_set_transcript(kind):
    assert kind in (VS_FS_SHA256, VS_FS_POSEIDON)
    _VS_FS[0] = kind


def _fs_embed(b):
    """bytes (<= 48) -> the Fq element whose Montgomery limbs are those bytes, zero-extended (no arithmetic to absorb)"""
    v = int.from_bytes(b, "little")
    return v % Q * inv(FQ_MONT_R, Q) % Q


def vs_fs_elements(cvk, proof):
    """The absorbed elements (73 for one instance; every further instance of a VarunaShaped-v2 proof with instances adds
    its 16 inputs as two limbs each and the coordinates of its two commitments after C_1, and its two evaluations at
    the end: 38 more), in order: tag, circuit digest, the 16 public inputs as two 128-bit limbs each (the
    nonnative absorb), the coordinates of C_0, C_1 | C_2..C_4 | C_5..C_8 | the 21 evaluations (the opening witnesses are
    made from xi and so come after it).  Two elements per permutation (the last block holds one): 19 + 3 + 4 + 11 = 37
    permutations; z_beta, z_gamma, xi are squeezed after the 22nd, 26th and 37th."""
    raw = proof["raw"]
    inputs, comms, evals = raw[16:528], raw[528:1392], raw[1392:2064]
    extra = proof.get("extra_raw", [])
    els = [_fs_embed(VS_TAG), _fs_embed(cvk["digest"])]
    parts = proof.get("parts_used", [])  # VarunaShaped-v3: the further circuits of the proof (header, digest | vk evaluations)
    for pr in parts:
        els += [_fs_embed(pr["raw"][:16]), _fs_embed(pr["digest"])]
    for k in range(VS_INPUTS):
        els += [_fs_embed(inputs[32 * k:32 * k + 16]), _fs_embed(inputs[32 * k + 16:32 * k + 32])]
    els += [_fs_embed(comms[48 * c:48 * c + 48]) for c in range(4)]
    for x in extra:  # 768 bytes: inputs, two commitments, two evaluations
        for k in range(VS_INPUTS):
            els += [_fs_embed(x[32 * k:32 * k + 16]), _fs_embed(x[32 * k + 16:32 * k + 32])]
        els += [_fs_embed(x[512 + 48 * c:512 + 48 * c + 48]) for c in range(4)]
    els += [_fs_embed(comms[48 * c:48 * c + 48]) for c in range(4, 18)]
    els += [_fs_embed(evals[32 * k:32 * k + 32]) for k in range(VS_EVALS)]
    for x in extra:
        els += [_fs_embed(x[704:736]), _fs_embed(x[736:768])]
    for pr in parts:
        els += [_fs_embed(pr["raw"][16 + 32 * k:48 + 32 * k]) for k in range(VS_VK_COMMS)]
    assert len(els) == 73 + 38 * len(extra) + 14 * len(parts)
    return els


def vs_challenges_poseidon(cvk, proof):
    els = vs_fs_elements(cvk, proof)
    m1 = len(proof.get("extra_raw", []))  # instances beyond the first
    d1 = len(proof.get("parts_used", []))  # circuits beyond the first
    st = [0, 0, 0]
    out = []
    for b in range(37 + 19 * m1 + 7 * d1):
        st[1] = (st[1] + els[2 * b]) % Q
        if 2 * b + 1 < len(els):
            st[2] = (st[2] + els[2 * b + 1]) % Q
        st = poseidon_permute(st)
        if b in (21 + 18 * m1 + d1, 25 + 18 * m1 + d1, 36 + 19 * m1 + 7 * d1):
            out.append(st[1] & ((1 << 128) - 1))  # low 128 bits of the canonical value of the first rate element
    return tuple(out)


def vs_proof_is_well_formed(proof):
    """what snarkVM's deserialiser guarantees and this ABI must check itself: every G1 point on the curve (or the (0,0)
    encoding of infinity), every coordinate < q, every Fr field < r (subgroup membership is the caller's precondition)"""
    raw = proof["raw"]
    for k in range(VS_INPUTS):
        if int.from_bytes(raw[16 + 32 * k:48 + 32 * k], "little") >= R:
            return False
    for k in range(VS_EVALS):
        if int.from_bytes(raw[1392 + 32 * k:1424 + 32 * k], "little") >= R:
            return False
    for off in [528 + 96 * i for i in range(VS_PROOF_COMMS)] + [2064, 2160]:
        xm, ym = int.from_bytes(raw[off:off + 48], "little"), int.from_bytes(raw[off + 48:off + 96], "little")
        if xm >= Q or ym >= Q:
            return False
        if xm == 0 and ym == 0:
            continue
        if not g1_is_on_curve((fq_from_mont(xm), fq_from_mont(ym))):
            return False
    return True


def vs_challenges(cvk, proof):
    """Transcript over the in-memory (Montgomery, little-endian) bytes of the proof."""
    if _VS_FS[0] == VS_FS_POSEIDON:
        return vs_challenges_poseidon(cvk, proof)
    raw = proof["raw"]
    inputs = raw[16:16 + 512]
    comms = raw[528:528 + 864]
    evals = raw[1392:1392 + 672]
    extra = proof.get("extra_raw", [])
    h = sha256(VS_TAG + cvk["digest"] + inputs + comms[:192])
    for pr in proof.get("parts_used", []):  # every further circuit: its header (id, first instance) and digest
        h = sha256(h + pr["raw"][:16] + pr["digest"])
    for x in extra:  # every further instance: its inputs and two commitments, chained
        h = sha256(h + x[:704])
    h = sha256(h + comms[192:480])
    zb = int.from_bytes(h[:16], "little")
    h = sha256(h + comms[480:864])
    zg = int.from_bytes(h[:16], "little")
    h = sha256(h + evals)
    for x in extra:  # and its two evaluations
        h = sha256(h + x[704:768])
    for pr in proof.get("parts_used", []):  # and the evaluations of every further circuit's polynomials
        h = sha256(h + pr["raw"][16:400])
    xi = int.from_bytes(h[:16], "little")
    return zb, zg, xi


def vs_query_of(p):
    if p < VS_PROOF_COMMS:
        return 0 if p < 5 else 1
    return 0 if (p - VS_PROOF_COMMS) < 6 else 1


def vs_verify_one(tau_h, circuits, proof, r_beta, r_gamma):
    """Per-proof verdict = public-input consistency AND the two KZG openings (randomised together)."""
    if proof["circuit"] >= len(circuits) or not vs_proof_is_well_formed(proof):
        return False
    cvk = circuits[proof["circuit"]]
    zb, zg, xi = vs_challenges(cvk, proof)
    z = (zb, zg)
    r = (r_beta, r_gamma)
    ev = proof["evals"]
    # algebraic check
    zn = pow(zb, VS_INPUTS, R)
    if zn == 1:
        return False
    w = omega(4)
    xhat = 0
    for k in range(VS_INPUTS):
        wk = pow(w, k, R)
        xhat += proof["inputs"][k] * wk * inv((zb - wk) % R, R)
    xhat = xhat * (zn - 1) * inv(VS_INPUTS, R) % R
    alg_ok = ev[0] == (xhat + (zn - 1) * ev[1]) % R
    total_c, total_w, gs = None, None, 0
    for p in range(VS_EVALS):
        q = vs_query_of(p)
        cm = proof["comms"][p] if p < VS_PROOF_COMMS else cvk["comms"][p - VS_PROOF_COMMS]
        coef = r[q] * pow(xi, p, R) % R
        total_c = g1_add(total_c, g1_mul(cm, coef))
        gs = (gs + coef * ev[p]) % R
    for q in range(2):
        total_c = g1_add(total_c, g1_mul(proof["w"][q], r[q] * z[q] % R))
        total_w = g1_add(total_w, g1_mul(proof["w"][q], r[q]))
    total_c = g1_add(total_c, g1_neg(g1_mul(G1_GEN, gs)))
    return alg_ok and pairing_product_is_one([(total_c, G2_GEN), (g1_neg(total_w), tau_h)])


# ----------------------------------------------------------------------------- VarunaShaped-v2 (rows A5 / A6)
VS2_PROOF_BYTES, VS2_MAX_BOUNDS = 3184, 2
VS1_QUERY_OF = [0] * 5 + [1] * 4 + [0] * 6 + [1] * 6  # the hard-wired split of v1 as a table


def vs2_parse_proof(b):
    assert len(b) == VS2_PROOF_BYTES
    d = vs_parse_proof(b[:VS_PROOF_BYTES])
    d["raw2"] = bytes(b)
    d["random_v"] = [_rd_fr(b[2256 + 32 * q:2288 + 32 * q]) for q in range(2)]
    d["shifted"] = [_rd_g1(b[2320 + 96 * j:2416 + 96 * j]) for j in range(VS_PROOF_COMMS)]
    return d


VS3_PROOF_BYTES, VS_MAX_INSTANCES, VS_INSTANCE_BYTES = 5504, 4, 768


def vs3_parse_proof(b):
    """mi_vs_proof3: a v2 record, the number of instances, and the further instances (inputs, A_j, B_j, a_j, b_j)"""
    assert len(b) == VS3_PROOF_BYTES
    d = vs2_parse_proof(b[:VS2_PROOF_BYTES])
    n = int.from_bytes(b[3184:3188], "little")
    d["n_instances"] = n
    d["extra_raw"] = [bytes(b[3200 + 768 * j:3200 + 768 * (j + 1)]) for j in range(max(0, min(n, VS_MAX_INSTANCES) - 1))]
    d["extra"] = [{"inputs": [_rd_fr(x[32 * k:32 * k + 32]) for k in range(VS_INPUTS)],
                   "comms": [_rd_g1(x[512 + 96 * c:608 + 96 * c]) for c in range(2)],
                   "evals": [_rd_fr(x[704 + 32 * k:736 + 32 * k]) for k in range(2)]} for x in d["extra_raw"]]
    return d


def vs2_verify_one(tau_h, gamma_g, shift_h, shape, circuits, proof, rs):
    """shape = {"query_of": [21], "bound_class": [9], "hiding": bool, "n_bounds": int}; rs = (r_beta, r_gamma, r_3).
    e(total_c, h) e(-total_w, tau h) prod_c e(bound_c, shift_h[c]) == 1 and the public-input identity."""
    if proof["circuit"] >= len(circuits) or not vs_proof_is_well_formed(proof):
        return False
    raw = proof["raw2"]
    for q in range(2):
        if int.from_bytes(raw[2256 + 32 * q:2288 + 32 * q], "little") >= R:
            return False
    for j in range(VS_PROOF_COMMS):
        xm = int.from_bytes(raw[2320 + 96 * j:2368 + 96 * j], "little")
        ym = int.from_bytes(raw[2368 + 96 * j:2416 + 96 * j], "little")
        if xm >= Q or ym >= Q or ((xm or ym) and not g1_is_on_curve((fq_from_mont(xm), fq_from_mont(ym)))):
            return False
    if "n_instances" in proof:  # a record with instances: 1 .. max_instances of the shape, each well-formed
        if not 1 <= proof["n_instances"] <= shape.get("max_instances", 1):
            return False
        for x in proof["extra_raw"]:
            for k in list(range(VS_INPUTS)) + [22, 23]:
                if int.from_bytes(x[32 * k:32 * k + 32], "little") >= R:
                    return False
            for c in range(2):
                xm = int.from_bytes(x[512 + 96 * c:560 + 96 * c], "little")
                ym = int.from_bytes(x[560 + 96 * c:608 + 96 * c], "little")
                if xm >= Q or ym >= Q or ((xm or ym) and not g1_is_on_curve((fq_from_mont(xm), fq_from_mont(ym)))):
                    return False
    cvk = circuits[proof["circuit"]]
    if "n_circuits" in proof:  # VarunaShaped-v3: the record names the further circuits of the proof
        if not vs4_circuits_ok(proof, shape, len(circuits)):
            return False
        for pr in proof["parts"]:
            for k in range(VS_VK_COMMS):
                if int.from_bytes(pr["raw"][16 + 32 * k:48 + 32 * k], "little") >= R:
                    return False
            pr["digest"] = circuits[pr["circuit"]]["digest"]
        proof["parts_used"] = proof["parts"]
    zb, zg, xi = vs_challenges(cvk, proof)
    z = (zb, zg)
    ev = proof["evals"]
    zn = pow(zb, VS_INPUTS, R)
    if zn == 1:
        return False
    lc_kinds = shape.get("lc_kind", [0] * VS_EVALS)
    if any(k & 4 for k in lc_kinds) and zb == zg:
        return False  # a coefficient 1 / (z_beta - z_gamma) does not exist
    inv_diff = inv((zb - zg) % R, R)

    def kappa(kind):
        k = pow((pow(zb, 16, R) - 1) % R, kind & 1, R) * pow((pow(zg, 16, R) - 1) % R, (kind >> 1) & 1, R) % R
        return k * inv_diff % R if kind & 4 else k

    w = omega(4)

    def input_identity(inputs, v0, v1):
        xhat = 0
        for k in range(VS_INPUTS):
            wk = pow(w, k, R)
            xhat += inputs[k] * wk * inv((zb - wk) % R, R)
        xhat = xhat * (zn - 1) * inv(VS_INPUTS, R) % R
        return v0 == (xhat + (zn - 1) * v1) % R

    alg_ok = input_identity(proof["inputs"], ev[0], ev[1])
    total_c, total_w, gs, hs = None, None, 0, 0
    # every further instance: its own public-input identity, and its two polynomials join the beta query with the
    # powers of xi after the 21 polynomials and the 9 degree-bound terms
    for j, inst in enumerate(proof.get("extra", [])):
        alg_ok = alg_ok and input_identity(inst["inputs"], inst["evals"][0], inst["evals"][1])
        for c in range(2):
            coef = rs[0] * pow(xi, VS_EVALS + VS_PROOF_COMMS + 2 * j + c, R) % R
            total_c = g1_add(total_c, g1_mul(inst["comms"][c], coef))
            gs = (gs + coef * inst["evals"][c]) % R
    bound = [None] * VS2_MAX_BOUNDS
    for p in range(VS_EVALS):
        q = shape["query_of"][p]
        cm = proof["comms"][p] if p < VS_PROOF_COMMS else cvk["comms"][p - VS_PROOF_COMMS]
        coef = rs[q] * pow(xi, p, R) * kappa(lc_kinds[p]) % R
        total_c = g1_add(total_c, g1_mul(cm, coef))
        gs = (gs + coef * ev[p]) % R
    # every further circuit of the proof: its 12 polynomials open like the first circuit's (same query and
    # linear-combination tables) with the powers of xi after the instances': xi^(36 + 12 (d - 1) + k)
    for d, pr in enumerate(proof.get("parts_used", [])):
        for k in range(VS_VK_COMMS):
            p = VS_PROOF_COMMS + k
            q = shape["query_of"][p]
            coef = rs[q] * pow(xi, VS4_FIRST_POWER + VS_VK_COMMS * d + k, R) * kappa(lc_kinds[p]) % R
            total_c = g1_add(total_c, g1_mul(circuits[pr["circuit"]]["comms"][k], coef))
            gs = (gs + coef * pr["vk_evals"][k]) % R
    for q in range(2):
        total_c = g1_add(total_c, g1_mul(proof["w"][q], rs[q] * z[q] % R))
        total_w = g1_add(total_w, g1_mul(proof["w"][q], rs[q]))
        if shape["hiding"]:
            hs = (hs + rs[q] * proof["random_v"][q]) % R
    total_c = g1_add(total_c, g1_neg(g1_mul(G1_GEN, gs)))
    if shape["hiding"]:
        total_c = g1_add(total_c, g1_neg(g1_mul(gamma_g, hs)))
    for j in range(VS_PROOF_COMMS):
        c = shape["bound_class"][j]
        if c < 0 or c >= shape["n_bounds"]:
            continue
        rho = rs[2] * pow(xi, VS_EVALS + j, R) % R
        total_c = g1_add(total_c, g1_neg(g1_mul(proof["shifted"][j], rho)))
        bound[c] = g1_add(bound[c], g1_mul(proof["comms"][j], rho))
    pairs = [(total_c, G2_GEN), (g1_neg(total_w), tau_h)] + [(bound[c], shift_h[c]) for c in range(shape["n_bounds"])]
    return alg_ok and pairing_product_is_one(pairs)


# ----------------------------------------------------------------------------- VarunaShaped-v3 (rows A7 + 8f-2)
# (a) proofs that cover instances of SEVERAL circuits (Varuna's keys_to_inputs: BTreeMap<vk, &[inputs]>): record
#     mi_vs_proof4 = a proof3 record, the number of distinct circuits and, for every circuit after the first, a part
#     {circuit id, index of its first instance, the evaluations of its 12 polynomials}.  Circuit ids ascend strictly
#     (a BTreeMap's key order), every circuit covers at least one instance; instance 0 belongs to the first circuit.
# (b) linear-combination coefficients that need the inverse of a challenge: lc_kind bit 2 = times 1 / (z_beta - z_gamma).
# (c) the WIRE form of a record (what the node holds inside Transaction bytes): every G1 point compressed to 48 bytes,
#     every Fr field canonical little-endian, headers as they are; slots the record does not use (shifted commitments of
#     unbounded polynomials, instances and circuits beyond the counts) are zero and are not decoded.
VS4_PROOF_BYTES, VS_MAX_CIRCUITS, VS_CIRCUIT_PART_BYTES, VS4_FIRST_POWER = 6320, 3, 400, 36


def vs4_parse_proof(b):
    """mi_vs_proof4: a proof3 record, n_circuits, and the further circuits' parts"""
    assert len(b) == VS4_PROOF_BYTES
    d = vs3_parse_proof(b[:VS3_PROOF_BYTES])
    nc = int.from_bytes(b[5504:5508], "little")
    d["n_circuits"] = nc
    d["parts"] = []
    for k in range(max(0, min(nc, VS_MAX_CIRCUITS) - 1)):
        raw = bytes(b[5520 + 400 * k:5520 + 400 * (k + 1)])
        d["parts"].append({"raw": raw, "circuit": int.from_bytes(raw[:4], "little"), "first": int.from_bytes(raw[4:8], "little"),
                           "vk_evals": [_rd_fr(raw[16 + 32 * j:48 + 32 * j]) for j in range(VS_VK_COMMS)]})
    return d


def vs4_circuits_ok(proof, shape, n_ctx_circuits):
    """1 <= n_circuits <= min(max_circuits, n_instances); circuit ids strictly ascending from the first circuit's and
    below the context's count; first-instance indices strictly ascending in [1, n_instances)"""
    nc, ni = proof["n_circuits"], proof["n_instances"]
    if not 1 <= nc <= min(shape.get("max_circuits", 1), VS_MAX_CIRCUITS) or nc > ni:
        return False
    c_prev, f_prev = proof["circuit"], 0
    for pr in proof["parts"]:
        if not (c_prev < pr["circuit"] < n_ctx_circuits and f_prev < pr["first"] < ni):
            return False
        c_prev, f_prev = pr["circuit"], pr["first"]
    return True


def vs_record_layout(kind):
    """segments of a record: ("raw", bytes) | ("fr", count) | ("g1", count, slot class); kind 1..4 = mi_vs_proof ..
    mi_vs_proof4.  Slot classes: "base" (always used), "shifted", ("inst", j), ("part", d)"""
    seg = [("raw", 16, "base"), ("fr", VS_INPUTS, "base"), ("g1", VS_PROOF_COMMS, "base"), ("fr", VS_EVALS, "base"), ("g1", 2, "base")]
    if kind >= 2:
        seg += [("fr", 2, "base"), ("g1", VS_PROOF_COMMS, "shifted")]
    if kind >= 3:
        seg += [("raw", 16, "base")]
        for j in range(VS_MAX_INSTANCES - 1):
            seg += [("fr", VS_INPUTS, ("inst", j)), ("g1", 2, ("inst", j)), ("fr", 2, ("inst", j))]
    if kind >= 4:
        seg += [("raw", 16, "base")]
        for d in range(VS_MAX_CIRCUITS - 1):
            seg += [("raw", 16, ("part", d)), ("fr", VS_VK_COMMS, ("part", d))]
    return seg


VS_RECORD_BYTES = {1: VS_PROOF_BYTES, 2: VS2_PROOF_BYTES, 3: VS3_PROOF_BYTES, 4: VS4_PROOF_BYTES}
VS_WIRE_BYTES = {1: 1728, 2: 2224, 3: 4256, 4: 5072}


def vs_record_kind(shape):
    if shape is None:
        return 1
    if shape.get("max_circuits", 1) > 1:
        return 4
    return 3 if shape.get("max_instances", 1) > 1 else 2


def _vs_slot_used(cls, j_in_seg, shape, m1, d1):
    if cls == "base":
        return True
    if cls == "shifted":
        c = shape["bound_class"][j_in_seg] if shape else -1
        return 0 <= c < shape["n_bounds"]
    if cls[0] == "inst":
        return cls[1] < m1
    return cls[1] < d1


def _vs_counts(rec_or_wire, kind, shape, wire):
    """(m1, d1): further instances / circuits the record's headers announce, clamped to what the shape allows (a count
    out of range leaves the first instance / circuit only, as in the uncompressed path)"""
    m1 = d1 = 0
    if kind >= 3:
        off = VS_WIRE_BYTES[2] if wire else VS2_PROOF_BYTES
        ni = int.from_bytes(rec_or_wire[off:off + 4], "little")
        if 1 <= ni <= min(shape.get("max_instances", 1), VS_MAX_INSTANCES):
            m1 = ni - 1
    if kind >= 4:
        off = VS_WIRE_BYTES[3] if wire else VS3_PROOF_BYTES
        nc = int.from_bytes(rec_or_wire[off:off + 4], "little")
        if 1 <= nc <= min(shape.get("max_circuits", 1), VS_MAX_CIRCUITS):
            d1 = nc - 1
    return m1, d1


def vs_wire_encode(record, shape=None):
    """uncompressed record (Montgomery fields, affine points) -> wire bytes; unused slots become zero"""
    kind = vs_record_kind(shape)
    assert len(record) == VS_RECORD_BYTES[kind]
    m1, d1 = _vs_counts(record, kind, shape or {}, False)
    out, o = bytearray(), 0
    for seg in vs_record_layout(kind):
        if seg[0] == "raw":
            out += record[o:o + seg[1]] if _vs_slot_used(seg[2], 0, shape, m1, d1) else bytes(seg[1])
            o += seg[1]
        elif seg[0] == "fr":
            for j in range(seg[1]):
                used = _vs_slot_used(seg[2], j, shape, m1, d1)
                out += fr_from_mont(int.from_bytes(record[o:o + 32], "little") % R).to_bytes(32, "little") if used else bytes(32)
                o += 32
        else:
            for j in range(seg[1]):
                used = _vs_slot_used(seg[2], j, shape, m1, d1)
                out += g1_compress(_rd_g1(record[o:o + 96])) if used else bytes(48)
                o += 96
    assert o == len(record) and len(out) == VS_WIRE_BYTES[kind]
    return bytes(out)


def vs_wire_decode(wire, shape=None):
    """wire bytes -> (uncompressed record, ok).  ok is False when a used point is not the encoding of a point of the
    order-r subgroup or a used Fr field is not below r; that field is then zero (the point: infinity) in the record."""
    kind = vs_record_kind(shape)
    assert len(wire) == VS_WIRE_BYTES[kind]
    m1, d1 = _vs_counts(wire, kind, shape or {}, True)
    out, o, ok = bytearray(), 0, True
    for seg in vs_record_layout(kind):
        if seg[0] == "raw":
            out += wire[o:o + seg[1]] if _vs_slot_used(seg[2], 0, shape, m1, d1) else bytes(seg[1])
            o += seg[1]
        elif seg[0] == "fr":
            for j in range(seg[1]):
                v = int.from_bytes(wire[o:o + 32], "little")
                o += 32
                if not _vs_slot_used(seg[2], j, shape, m1, d1):
                    out += bytes(32)
                elif v >= R:
                    ok = False
                    out += bytes(32)
                else:
                    out += fr_to_mont(v).to_bytes(32, "little")
        else:
            for j in range(seg[1]):
                enc = wire[o:o + 48]
                o += 48
                if not _vs_slot_used(seg[2], j, shape, m1, d1):
                    out += bytes(96)
                    continue
                good, pt = g1_decompress(enc)
                if not good:
                    ok = False
                if not good or pt is None:
                    out += bytes(96)
                else:
                    out += fq_to_mont(pt[0]).to_bytes(48, "little") + fq_to_mont(pt[1]).to_bytes(48, "little")
    assert o == len(wire) and len(out) == VS_RECORD_BYTES[kind]
    return bytes(out), ok


def vs_parse_record(b, shape=None):
    return {1: vs_parse_proof, 2: vs2_parse_proof, 3: vs3_parse_proof, 4: vs4_parse_proof}[vs_record_kind(shape)](b)


# ----------------------------------------------------------------------------- Poseidon sponge over Fq (row A9)
# Structure as recalled for snarkVM's PoseidonSponge<Fq, RATE = 2, CAPACITY = 1>: width 3, S-box x^17, 8 full +
# 31 partial rounds.  snarkVM's round constants and MDS matrix are NOT available offline, so the parameters here
# are generated by a documented procedure of our own (SHA-256 counter stream, Cauchy matrix): same shape and cost,
# different numbers -> parity unpinned, like the rest of the path.
POSEIDON_T, POSEIDON_RF, POSEIDON_RP, POSEIDON_ALPHA = 3, 8, 31, 17
POSEIDON_TAG = b"MI355X-Poseidon-BLS12-377-Fq-v1"


def poseidon_params():
    rounds = POSEIDON_RF + POSEIDON_RP
    ark = []
    for r in range(rounds):
        row = []
        for i in range(POSEIDON_T):
            seed = POSEIDON_TAG + bytes([r, i])
            v = int.from_bytes(sha256(seed + b"\x00") + sha256(seed + b"\x01"), "little") % Q
            row.append(v)
        ark.append(row)
    mds = [[inv((i + POSEIDON_T + j) % Q, Q) for j in range(POSEIDON_T)] for i in range(POSEIDON_T)]  # 1/(x_i + y_j), x_i = i, y_j = T + j
    return ark, mds


def poseidon_permute(state, params=None):
    ark, mds = params or poseidon_params()
    st = list(state)
    half = POSEIDON_RF // 2
    for r in range(POSEIDON_RF + POSEIDON_RP):
        st = [(x + c) % Q for x, c in zip(st, ark[r])]
        if r < half or r >= half + POSEIDON_RP:
            st = [pow(x, POSEIDON_ALPHA, Q) for x in st]
        else:
            st[0] = pow(st[0], POSEIDON_ALPHA, Q)
        st = [sum(mds[i][j] * st[j] for j in range(POSEIDON_T)) % Q for i in range(POSEIDON_T)]
    return st


def poseidon_hash(inputs, n_out, params=None):
    """Sponge: state = [capacity, rate0, rate1]; absorb two elements per permutation (the last block may hold one),
    one permutation after the last block, then squeeze rate0, rate1, permute, ..."""
    params = params or poseidon_params()
    st = [0, 0, 0]
    for i in range(0, len(inputs), 2):
        st[1] = (st[1] + inputs[i]) % Q
        if i + 1 < len(inputs):
            st[2] = (st[2] + inputs[i + 1]) % Q
        st = poseidon_permute(st, params)
    if not inputs:
        st = poseidon_permute(st, params)
    out = []
    while len(out) < n_out:
        out.append(st[1])
        if len(out) < n_out:
            out.append(st[2])
        if len(out) < n_out:
            st = poseidon_permute(st, params)
    return out


# ----------------------------------------------------------------------------- Schnorr signatures over Edwards-BLS12
# What the BFT layer verifies per batch certificate: reference node/bft/src/helpers/storage.rs:465 (check_certificate ->
# every signer's signature over the batch id), node/bft/src/gateway.rs:1391 (handshake, verify_bytes), account/src/lib.rs:76-88.
# The scheme is snarkVM's console::account::Signature [RECALL: the crate is not in /root/reference; PARITY UNPINNED]:
#   curve      twisted Edwards  a x^2 + y^2 = 1 + d x^2 y^2  over Fr of BLS12-377, a = -1, d = 3021; #E = 4 s with s the
#              251-bit prime below (machine-checked here: s prime, 4 s in the Hasse interval, 4 s P = O for random P;
#              a is a square and d is not, so the addition law is complete)
#   keys       pk_sig = sk_sig G, pr_sig = r_sig G, sk_prf = H4(pk_sig.x, pr_sig.x), address = pk_sig + pr_sig + sk_prf G
#   signature  (challenge c, response s, pk_sig, pr_sig) with s = k - c sk_sig for a nonce k, c = H8(kG.x, pk_sig.x, pr_sig.x,
#              address.x, message...)
#   verify     g_r = s G + c pk_sig;  c == H8(g_r.x, pk_sig.x, pr_sig.x, address.x, message...)  and the address identity.
# H4 / H8: hash-to-scalar by a Poseidon sponge over Fr of rate 4 / 8 (width 5 / 9, x^17, 8 + 31 rounds), one permutation
# for these input counts, the first rate element truncated to 250 bits.  snarkVM's generator, Poseidon constants and domain
# separators are not available offline: the generator is derived below, the constants by the procedure of poseidon_params
# over Fr -- same shape and cost, different numbers.
ED_A, ED_D = R - 1, 3021
ED_S = 2111115437357092606062206234695386632838870926408408195193685246394721360383  # subgroup order (251 bits); cofactor 4
ED_ID = (0, 1)


def ed_on_curve(p):
    x, y = p
    return (ED_A * x * x + y * y - 1 - ED_D * x * x * y * y) % R == 0


def ed_add(p, q):
    x1, y1 = p
    x2, y2 = q
    t = ED_D * x1 * x2 * y1 * y2 % R
    return ((x1 * y2 + y1 * x2) * inv((1 + t) % R, R) % R, (y1 * y2 - ED_A * x1 * x2) * inv((1 - t) % R, R) % R)


def ed_mul(p, k):
    acc = ED_ID
    for bit in bin(k)[2:] if k else "":
        acc = ed_add(acc, acc)
        if bit == "1":
            acc = ed_add(acc, p)
    return acc


def fr_sqrt(v):
    """a square root of v mod R (Tonelli-Shanks over the 2^47-th roots of unity), or None"""
    v %= R
    if v == 0:
        return 0
    if pow(v, (R - 1) // 2, R) != 1:
        return None
    s, t = FR_TWO_ADICITY, (R - 1) >> FR_TWO_ADICITY
    m, c, x, b = s, pow(FR_GENERATOR, t, R), pow(v, (t + 1) // 2, R), pow(v, t, R)
    while b != 1:
        i, b2 = 0, b
        while b2 != 1:
            b2, i = b2 * b2 % R, i + 1
        g = pow(c, 1 << (m - i - 1), R)
        m, c, x, b = i, g * g % R, x * g % R, b * g * g % R
    return x


def ed_generator():
    """4 (x, y) for the smallest x >= 2 that is the abscissa of a curve point, y the smaller root: a generator of the
    order-s subgroup (own constant; snarkVM's is not available offline)"""
    x = 2
    while True:
        y = fr_sqrt((1 - ED_A * x * x) * inv((1 - ED_D * x * x) % R, R) % R)
        if y is not None:
            g = ed_mul((x, min(y, R - y)), 4)
            if g != ED_ID:
                assert ed_mul(g, ED_S) == ED_ID
                return g
        x += 1


ED_POSEIDON_TAG = b"MI355X-Poseidon-BLS12-377-Fr-v1"


def ed_poseidon_params(width):
    rounds = POSEIDON_RF + POSEIDON_RP
    ark = [[int.from_bytes(sha256(ED_POSEIDON_TAG + bytes([width, r, i]) + b"\x00") + sha256(ED_POSEIDON_TAG + bytes([width, r, i]) + b"\x01"), "little") % R
            for i in range(width)] for r in range(rounds)]
    mds = [[inv((i + width + j) % R, R) for j in range(width)] for i in range(width)]
    return ark, mds


_ED_PARAMS = {}


def ed_poseidon_permute(state):
    w = len(state)
    if w not in _ED_PARAMS:
        _ED_PARAMS[w] = ed_poseidon_params(w)
    ark, mds = _ED_PARAMS[w]
    st = list(state)
    half = POSEIDON_RF // 2
    for r in range(POSEIDON_RF + POSEIDON_RP):
        st = [(x + c) % R for x, c in zip(st, ark[r])]
        if r < half or r >= half + POSEIDON_RP:
            st = [pow(x, POSEIDON_ALPHA, R) for x in st]
        else:
            st[0] = pow(st[0], POSEIDON_ALPHA, R)
        st = [sum(mds[i][j] * st[j] for j in range(w)) % R for i in range(w)]
    return st


def ed_hash_to_scalar(inputs, rate):
    """one permutation of the width rate + 1 sponge: capacity = 2^64 * rate + number of inputs (domain separation), the
    inputs in the rate part; the first rate element's low 250 bits"""
    assert len(inputs) <= rate
    st = [((rate << 64) + len(inputs)) % R] + [x % R for x in inputs] + [0] * (rate - len(inputs))
    return ed_poseidon_permute(st)[1] & ((1 << 250) - 1)


def schnorr_keygen(sk_sig, r_sig, g=None):
    g = g or ed_generator()
    pk_sig, pr_sig = ed_mul(g, sk_sig % ED_S), ed_mul(g, r_sig % ED_S)
    sk_prf = ed_hash_to_scalar([pk_sig[0], pr_sig[0]], 4)
    return pk_sig, pr_sig, ed_add(ed_add(pk_sig, pr_sig), ed_mul(g, sk_prf))


def schnorr_sign(sk_sig, r_sig, nonce, message, g=None):
    g = g or ed_generator()
    pk_sig, pr_sig, address = schnorr_keygen(sk_sig, r_sig, g)
    g_r = ed_mul(g, nonce % ED_S)
    c = ed_hash_to_scalar([g_r[0], pk_sig[0], pr_sig[0], address[0]] + list(message), 8)
    return {"challenge": c, "response": (nonce - c * sk_sig) % ED_S, "pk_sig": pk_sig, "pr_sig": pr_sig}, address


def schnorr_verify(address, message, sig, g=None):
    g = g or ed_generator()
    c, s, pk_sig, pr_sig = sig["challenge"], sig["response"], sig["pk_sig"], sig["pr_sig"]
    if c >= ED_S or s >= ED_S or not (ed_on_curve(pk_sig) and ed_on_curve(pr_sig) and ed_on_curve(address)):
        return False
    g_r = ed_add(ed_mul(g, s), ed_mul(pk_sig, c))
    if ed_hash_to_scalar([g_r[0], pk_sig[0], pr_sig[0], address[0]] + list(message), 8) != c:
        return False
    sk_prf = ed_hash_to_scalar([pk_sig[0], pr_sig[0]], 4)
    return ed_add(ed_add(pk_sig, pr_sig), ed_mul(g, sk_prf)) == address
