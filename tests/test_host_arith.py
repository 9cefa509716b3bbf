"""CPU tier: the product's shared host/device arithmetic (snarkos_amd/csrc/{field,curve,tower}.hpp),
compiled for the host with g++, pinned bit-exact against the CPU oracle.  The same source is what the
HIP kernels inline, so this catches arithmetic bugs without a GPU; the -m gpu tests then pin the device
build itself."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import pyref as P
from helpers import rng_points, rng_scalars

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hc():
    src = os.path.join(HERE, "hostlib", "host_check.cpp")
    so = os.path.join(HERE, "hostlib", "libhostcheck.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
    l = C.CDLL(so)
    l.hc_constants_ok.restype = C.c_int
    return l


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_limb_constants_match_generated(hc):
    assert hc.hc_constants_ok() == 0


def test_fq_ops(hc, orc):
    pts = rng_points(1, 64).reshape(-1, 6)
    special = [np.zeros(6, dtype=np.uint64), orc.fq_from_int(1), orc.fq_from_int(P.Q - 1), orc.int_to_limbs(P.Q - 1, 6)]
    vals = list(pts) + special
    out, want = np.zeros(6, dtype=np.uint64), np.zeros(6, dtype=np.uint64)
    for i in range(len(vals)):
        a = np.ascontiguousarray(vals[i]); b = np.ascontiguousarray(vals[(i * 7 + 3) % len(vals)])
        hc.hc_fq_mul32(_p(a), _p(b), _p(out)); orc.lib().orc_fq_mul(_p(want), _p(a), _p(b))
        assert (out == want).all(), ("mul32 (the device limb algorithm)", i)
        for name in ("mul", "add", "sub"):
            getattr(hc, "hc_fq_" + name)(_p(a), _p(b), _p(out))
            getattr(orc.lib(), "orc_fq_" + name)(_p(want), _p(a), _p(b))
            assert (out == want).all(), (name, i)
    for a in vals[:8] + special[:3]:
        a = np.ascontiguousarray(a)
        hc.hc_fq_inv(_p(a), _p(out)); orc.lib().orc_fq_inv(_p(want), _p(a))
        assert (out == want).all()
        hc.hc_fq_inv_egcd(_p(a), _p(out))
        assert (out == want).all()
        hc.hc_fq_inv_bingcd(_p(a), _p(out))
        assert (out == want).all()


def test_fast_inversion_many_inputs(hc, orc):
    """inv_bingcd (Pornin's approximated binary GCD, what invq / invr run on the device and on the host) against the
    Fermat inverse of the oracle: random values, values of every length (short values exercise the exact small-operand
    branch and the 63 / 64-bit window), p - 1, p - 2, 1, 2, 0"""
    g = np.random.default_rng(77)
    out, want = np.zeros(6, dtype=np.uint64), np.zeros(6, dtype=np.uint64)
    ints = [0, 1, 2, 3, P.Q - 1, P.Q - 2, (P.Q + 1) // 2, P.FQ_MONT_R]
    ints += [(1 << k) % P.Q for k in range(1, 384, 7)] + [((1 << k) - 1) % P.Q for k in (30, 31, 32, 33, 61, 62, 63, 64, 65, 95, 96, 97)]
    ints += [int(g.integers(1, 1 << 62)) for _ in range(20)]
    ints += [int.from_bytes(g.bytes(48), "little") % P.Q for _ in range(300)]
    for x in ints:
        a = np.ascontiguousarray(orc.int_to_limbs(x, 6))  # any residue is a valid Montgomery representative
        hc.hc_fq_inv_bingcd(_p(a), _p(out)); orc.lib().orc_fq_inv(_p(want), _p(a))
        assert (out == want).all(), hex(x)
    out, want = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
    ints = [0, 1, 2, P.R - 1, P.R - 2, (P.R + 1) // 2] + [(1 << k) % P.R for k in range(1, 256, 5)]
    ints += [int.from_bytes(g.bytes(32), "little") % P.R for _ in range(300)]
    for x in ints:
        a = np.ascontiguousarray(orc.int_to_limbs(x, 4))
        hc.hc_fr_inv_bingcd(_p(a), _p(out)); orc.lib().orc_fr_inv(_p(want), _p(a))
        assert (out == want).all(), hex(x)


def test_fr_ops(hc, orc):
    a = rng_scalars(2, 64); b = rng_scalars(3, 64)
    out, want = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
    for i in range(64):
        hc.hc_fr_mul(_p(a[i]), _p(b[i]), _p(out)); orc.lib().orc_fr_mul(_p(want), _p(a[i]), _p(b[i]))
        assert (out == want).all()
        hc.hc_fr_mul32(_p(a[i]), _p(b[i]), _p(out))
        assert (out == want).all()
    for i in range(4):
        hc.hc_fr_inv(_p(a[i]), _p(out)); orc.lib().orc_fr_inv(_p(want), _p(a[i]))
        assert (out == want).all()
        hc.hc_fr_inv_egcd(_p(a[i]), _p(out))
        assert (out == want).all()
        hc.hc_fr_inv_bingcd(_p(a[i]), _p(out))
        assert (out == want).all()
        hc.hc_fr_to_mont(_p(a[i]), _p(out)); orc.lib().orc_fr_from_canonical(_p(want), _p(a[i]))
        assert (out == want).all()
        hc.hc_fr_from_mont(_p(a[i]), _p(out)); orc.lib().orc_fr_to_canonical(_p(want), _p(a[i]))
        assert (out == want).all()


def test_g1_group_law(hc, orc):
    a, b = rng_points(4, 24), rng_points(5, 24)
    b[0] = a[0]
    b[1] = a[1]; b[1, 6:] = orc.fq_from_int((-orc.fq_to_int(a[1, 6:])) % P.Q)
    a[2] = 0
    b[3] = 0
    out = np.zeros(12, dtype=np.uint64)
    for i in range(24):
        pa, pb = orc.g1_to_ints(a[i]), orc.g1_to_ints(b[i])
        want = P.g1_add(pa, pb)
        for fn in (hc.hc_g1_add_mixed, hc.hc_g1_add):
            fn(_p(np.ascontiguousarray(a[i])), _p(np.ascontiguousarray(b[i])), _p(out))
            assert orc.g1_to_ints(out) == want, i
    hc.hc_g1_dbl(_p(np.ascontiguousarray(a[5])), _p(out))
    assert orc.g1_to_ints(out) == P.g1_add(orc.g1_to_ints(a[5]), orc.g1_to_ints(a[5]))
    k = rng_scalars(6, 1)[0]
    k32 = np.frombuffer(k.tobytes(), dtype=np.uint32).copy()
    hc.hc_g1_mul(_p(np.ascontiguousarray(a[6])), _p(k32), C.c_int(253), _p(out))
    assert (out == orc.g1_mul(a[6], orc.limbs_to_int(k))).all()


def test_host_fold_arithmetic_without_adx(orc):
    """the same checks through the portable product (MI_SNARK_HOSTFOLD_NO_ADX=1 is read once per process: a child)"""
    import subprocess, sys
    code = ("import os, sys; sys.path[:0] = [%r, %r]; import conftest, test_host_arith as t, orc_py; "
            "hc = t.hc.__wrapped__(); assert hc.hc_hf_has_adx() == 0; t.test_host_fold_arithmetic(hc, orc_py); print('ok')"
            % (HERE, os.path.join(os.path.dirname(HERE), "oracle")))
    env = dict(os.environ, MI_SNARK_HOSTFOLD_NO_ADX="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_host_fold_arithmetic(hc, orc):
    """csrc/hostfold.hpp (the host fold of a small MSM's per-bit sums): its Fq product / sum / difference on 64-bit
    limbs against the oracle, and the fold itself -- Horner over per-bit sums, split into a low part and a shifted high
    part (Jacobian doublings) -- against big-integer arithmetic"""
    rs = np.random.RandomState(99)
    vals = [orc.fq_from_int(int.from_bytes(rs.bytes(48), "little") % P.Q) for _ in range(200)]
    vals += [np.zeros(6, dtype=np.uint64), orc.fq_from_int(1), orc.fq_from_int(P.Q - 1), orc.fq_from_int(P.Q - 2)]
    out, want = np.zeros(6, dtype=np.uint64), np.zeros(6, dtype=np.uint64)
    for i in range(len(vals)):
        a = np.ascontiguousarray(vals[i]); b = np.ascontiguousarray(vals[(i * 5 + 1) % len(vals)] if i % 9 else vals[i])
        for name in ("mul", "add", "sub"):
            getattr(hc, "hc_hf_" + name)(_p(a), _p(b), _p(out))
            getattr(orc.lib(), "orc_fq_" + name)(_p(want), _p(a), _p(b))
            assert (out == want).all(), (name, i)
    assert hc.hc_hf_has_adx() in (0, 1)
    pts = rng_points(17, 40)
    pts[3] = 0; pts[11] = 0; pts[39] = 0   # empty bit positions, the top one too
    pts[7] = pts[6]                        # equal neighbours: 2 S + S
    total = None
    for g in range(39, -1, -1):
        total = P.g1_add(total, total) if total is not None else None
        pg = orc.g1_to_ints(pts[g])
        if pg is not None:
            total = P.g1_add(total, pg) if total is not None else pg
    o = np.zeros(12, dtype=np.uint64)
    for cut in (0, 1, 13, 39, 40):
        hc.hc_hf_fold(_p(np.ascontiguousarray(pts)), C.c_int(40), C.c_int(cut), _p(o))
        assert orc.g1_to_ints(o) == total, cut
    # nothing but empty positions
    hc.hc_hf_fold(_p(np.zeros((5, 12), dtype=np.uint64)), C.c_int(5), C.c_int(2), _p(o))
    assert not o.any()


def test_glv_scalar_multiplication(hc, orc):
    """glv.hpp, the per-proof fallback's scalar multiplication: s = s1 + lambda s2 with both halves below 2^127, signed
    window digits that recompose the value, and the whole chain (endomorphism, Jacobian doubling / mixed addition)
    against the oracle's double-and-add -- edge scalars included"""
    import random
    lam = P.X * P.X - 1
    assert (lam * lam + lam + 1) == P.R
    rnd = random.Random(11)
    edge = [0, 1, 2, lam - 1, lam, lam + 1, 2 * lam, P.R - 1, P.R - 2, (1 << 128) - 1, 1 << 127, (1 << 252) + 12345, lam * lam, lam * (lam + 1)]
    scalars = edge + [rnd.randrange(P.R) for _ in range(300)]
    s1, s2 = np.zeros(4, dtype=np.uint32), np.zeros(4, dtype=np.uint32)
    for s in scalars:
        s8 = np.frombuffer(orc.int_to_limbs(s, 4).tobytes(), dtype=np.uint32).copy()
        hc.hc_glv_split(_p(s8), _p(s1), _p(s2))
        a, b = orc.limbs_to_int(s1.view(np.uint64)), orc.limbs_to_int(s2.view(np.uint64))
        assert a + lam * b == s and a < lam and b < (1 << 127), hex(s)
    for w, nw in ((3, 43), (4, 33)):
        digs = np.zeros(nw, dtype=np.int32)
        for v in [0, 1, (1 << 128) - 1, (1 << 127) - 1, 1 << 127, int("4" * 32, 16), int("c" * 32, 16)] + [rnd.getrandbits(128) for _ in range(200)]:
            v4 = np.frombuffer(orc.int_to_limbs(v, 2).tobytes(), dtype=np.uint32).copy()
            hc.hc_glv_digits(C.c_int(w), _p(v4), _p(digs))
            assert sum(int(d) << (w * i) for i, d in enumerate(digs)) == v
            assert all(-(1 << (w - 1)) <= d <= (1 << (w - 1)) for d in digs)
    pts = rng_points(21, 6)
    pts[5] = 0
    out = np.zeros(12, dtype=np.uint64)
    for i, s in enumerate(edge + [rnd.randrange(P.R) for _ in range(10)]):
        s8 = np.frombuffer(orc.int_to_limbs(s, 4).tobytes(), dtype=np.uint32).copy()
        pt = np.ascontiguousarray(pts[i % 6])
        for w in (3, 4):
            hc.hc_glv_mul(C.c_int(w), _p(pt), _p(s8), _p(out))
            assert (out == orc.g1_mul(pt, s)).all(), (i, w)


def test_fq12_and_pairing(hc, orc):
    g, h = orc.g1_generator(), orc.g2_generator()
    p = orc.g1_mul(g, 0xABCDEF12345); q = orc.g2_mul(h, 0x13579BDF)
    e = orc.pairing(p, q)
    out = np.zeros(72, dtype=np.uint64)
    ps = np.ascontiguousarray(p); qs = np.ascontiguousarray(q)
    hc.hc_pairing_product(_p(ps), _p(qs), C.c_int(1), _p(out))
    assert (out == e).all()
    # tower mul / sqr / inv against the oracle
    e2 = orc.pairing(g, h)
    want = np.zeros(72, dtype=np.uint64)
    hc.hc_fq12_mul(_p(e), _p(e2), _p(out)); orc.lib().orc_fq12_mul(_p(want), _p(e), _p(e2))
    assert (out == want).all()
    hc.hc_fq12_sqr(_p(e), _p(out)); orc.lib().orc_fq12_mul(_p(want), _p(e), _p(e))
    assert (out == want).all()
    hc.hc_fq12_inv(_p(e), _p(out)); orc.lib().orc_fq12_inv(_p(want), _p(e))
    assert (out == want).all()
    # product with a cancelling pair is one
    a, b = 1234577, 99991
    ps = np.stack([orc.g1_mul(g, a), orc.g1_mul(g, (P.R - a * b % P.R) % P.R)])
    qs = np.stack([orc.g2_mul(h, b), h])
    hc.hc_pairing_product(_p(ps), _p(qs), C.c_int(2), _p(out))
    one = np.zeros(72, dtype=np.uint64); one[:6] = orc.int_to_limbs(P.FQ_MONT_R, 6)
    assert (out == one).all()


def test_compressed_points_host_build(hc, orc):
    """csrc/codec.hpp compiled for the host: the square root, the encoding both ways and the endomorphism subgroup
    test (u (u P) == -phi^2(P)) against the oracle's definition (r P == O) on the golden vectors -- subgroup points,
    the other root, a curve point outside the subgroup, non-residues -- and on random curve points."""
    import ctypes as C
    import golden_util as G
    import pyref as P
    d = G.load("g1_compressed.json")
    for c in d["points"]:
        raw = np.frombuffer(bytes.fromhex(c["bytes"]), dtype=np.uint8).copy()
        for check, key in ((1, "ok"), (0, "ok_unchecked")):
            out = np.zeros(12, dtype=np.uint64)
            ok = hc.hc_g1_decompress(orc._p(raw), C.c_int(check), orc._p(out))
            assert bool(ok) == c[key]
            assert orc.g1_to_ints(out) == (G.pt(c["point"]) if c[key] else None)
        if c["ok"]:
            back = np.zeros(48, dtype=np.uint8)
            hc.hc_g1_compress(orc._p(orc.g1_from_ints(G.pt(c["point"]))), orc._p(back))
            assert bytes(back).hex() == c["bytes"]
    rng = P.SplitMix64(0xC0DEC)
    in_sub = 0
    for k in range(40):
        x = rng.below(P.Q, k)
        y = P.fq_sqrt(x * x * x + 1)
        root = np.zeros(6, dtype=np.uint64)
        got = hc.hc_fq_sqrt(orc._p(orc.fq_from_int((x * x * x + 1) % P.Q)), orc._p(root))
        assert bool(got) == (y is not None)
        if y is None:
            continue
        assert orc.fq_to_int(root) in (y, P.Q - y)
        pt = (x, y) if k % 2 else P.g1_mul(P.G1_GEN, x % P.R)  # a random curve point, or a subgroup point
        want = P._g1_mul_any(pt, P.R) is None
        assert bool(hc.hc_g1_in_subgroup(orc._p(orc.g1_from_ints(pt)))) == want
        in_sub += want
    assert 0 < in_sub < 40
