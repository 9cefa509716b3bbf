#!/usr/bin/env python3
"""Generate tests/golden/*.json from oracle/pyref.py (independent Python big-ints), NOT from the C oracle.

There is no reference-side golden material for this path (parity unpinned), so the
committed vectors are seeded inputs -> outputs of the textbook definitions in pyref.py.  The C oracle
and (on the GPU box) the HIP library are both checked against these files.
Run:  python oracle/gen_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import pyref as P  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def hx(v):
    return "%x" % v


def pt(p):
    return None if p is None else [hx(p[0]), hx(p[1])]


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = P.SplitMix64(0xA1E00001)
    # --- field products
    fq = [[hx(a), hx(b), hx(a * b % P.Q)] for a, b in ((rng.below(P.Q, 7), rng.below(P.Q, 7)) for _ in range(16))]
    fr = [[hx(a), hx(b), hx(a * b % P.R)] for a, b in ((rng.below(P.R, 5), rng.below(P.R, 5)) for _ in range(16))]
    json.dump({"fq_mul": fq, "fr_mul": fr, "fq_mont_r": hx(P.FQ_MONT_R), "fr_mont_r": hx(P.FR_MONT_R)},
              open(os.path.join(OUT, "field.json"), "w"), indent=0)
    # --- G1 scalar multiples and an MSM with edge cases
    ks = [1, 2, 3, P.R - 1, rng.below(P.R, 5), rng.below(P.R, 5)]
    g1 = [[hx(k), pt(P.g1_mul(P.G1_GEN, k))] for k in ks]
    n = 48
    pk = [rng.below(P.R, 5) for _ in range(n)]
    pts = [P.g1_mul(P.G1_GEN, k) for k in pk]
    sc = [rng.below(P.R, 5) for _ in range(n)]
    sc[0] = 0
    sc[1] = P.R - 1
    sc[2] = 1
    pts[3] = None
    pts[5], sc[5] = pts[4], sc[4]
    pts[7], sc[7] = P.g1_neg(pts[6]), sc[6]
    msm = {"points": [pt(p) for p in pts], "scalars": [hx(s) for s in sc], "result": pt(P.msm(pts, sc)),
           "result_first_20": pt(P.msm(pts[:20], sc[:20]))}
    json.dump({"g1_mul": g1, "msm": msm}, open(os.path.join(OUT, "g1_msm.json"), "w"), indent=0)
    # --- NTT, 2^6, all four flavours
    a = [rng.below(P.R, 5) for _ in range(64)]
    ntt = {"input": [hx(x) for x in a], "omega_64": hx(P.omega(6)),
           "fft": [hx(x) for x in P.ntt(a)], "ifft": [hx(x) for x in P.ntt(a, inverse=True)],
           "coset_fft": [hx(x) for x in P.ntt(a, coset=True)],
           "coset_ifft": [hx(x) for x in P.ntt(a, inverse=True, coset=True)]}
    assert P.ntt(a) == P.dft(a, P.omega(6))
    json.dump(ntt, open(os.path.join(OUT, "ntt.json"), "w"), indent=0)
    # --- pairing value (cube of the reduced ate pairing) in the power basis Fq[w]/(w^12+5)
    ka, kb = rng.below(P.R, 5), rng.below(P.R, 5)
    p, q = P.g1_mul(P.G1_GEN, ka), P.g2_mul(P.G2_GEN, kb)
    e = P.pairing(p, q, cubed=True)
    pairing = {"a": hx(ka), "b": hx(kb), "p": pt(p), "q": [[hx(q[0][0]), hx(q[0][1])], [hx(q[1][0]), hx(q[1][1])]],
               "e_cubed_power_basis": [hx(c) for c in e]}
    json.dump(pairing, open(os.path.join(OUT, "pairing.json"), "w"), indent=0)
    # --- KZG batch instance with a known tau
    tau = rng.below(P.R, 5)
    polys = [[rng.below(P.R, 5) for _ in range(4)] for _ in range(5)]
    ev = lambda c, z: sum(ci * pow(z, i, P.R) for i, ci in enumerate(c)) % P.R  # noqa: E731
    zs = [rng.below(P.R, 5) for _ in range(5)]
    vs = [ev(c, z) for c, z in zip(polys, zs)]
    cs = [P.g1_mul(P.G1_GEN, ev(c, tau)) for c in polys]
    ws = [P.g1_mul(P.G1_GEN, (ev(c, tau) - v) * P.inv((tau - z) % P.R, P.R) % P.R) for c, z, v in zip(polys, zs, vs)]
    rs = [rng.below(1 << 128, 2) for _ in range(5)]
    tau_h = P.g2_mul(P.G2_GEN, tau)
    assert P.kzg_batch_check(tau_h, cs, zs, vs, ws, rs)
    bad = list(vs)
    bad[2] = (bad[2] + 1) % P.R
    assert not P.kzg_batch_check(tau_h, cs, zs, bad, ws, rs)
    kzg = {"tau": hx(tau), "tau_h": [[hx(tau_h[0][0]), hx(tau_h[0][1])], [hx(tau_h[1][0]), hx(tau_h[1][1])]],
           "comms": [pt(c) for c in cs], "points": [hx(z) for z in zs], "values": [hx(v) for v in vs],
           "bad_values": [hx(v) for v in bad], "witnesses": [pt(w) for w in ws], "randomizers": [hx(r) for r in rs],
           "verdict_good": True, "verdict_bad": False}
    json.dump(kzg, open(os.path.join(OUT, "kzg.json"), "w"), indent=0)
    # --- Poseidon sponge over Fq (own parameters, see pyref.poseidon_params)
    cases = []
    for m, k in ((0, 1), (1, 1), (2, 2), (5, 3)):
        ins = [rng.below(P.Q, 7) for _ in range(m)]
        cases.append({"inputs": [hx(x) for x in ins], "outputs": [hx(x) for x in P.poseidon_hash(ins, k)]})
    perm = P.poseidon_permute([1, 2, 3])
    json.dump({"permute_1_2_3": [hx(x) for x in perm], "sponge": cases}, open(os.path.join(OUT, "poseidon.json"), "w"), indent=0)
    # --- KZG batch instance WITH hiding (gamma_g, random_v): commitments C = p(tau) G + rho(tau) gamma_g
    gam = rng.below(P.R, 5)
    gamma_g = P.g1_mul(P.G1_GEN, gam)
    rhos = [[rng.below(P.R, 5) for _ in range(3)] for _ in range(5)]
    hcs = [P.g1_mul(P.G1_GEN, (ev(c, tau) + gam * ev(rh, tau)) % P.R) for c, rh in zip(polys, rhos)]
    rbars = [ev(rh, z) for rh, z in zip(rhos, zs)]
    hws = [P.g1_mul(P.G1_GEN, ((ev(c, tau) - v) + gam * (ev(rh, tau) - rb)) * P.inv((tau - z) % P.R, P.R) % P.R)
           for c, rh, z, v, rb in zip(polys, rhos, zs, vs, rbars)]
    assert P.kzg_batch_check_hiding(tau_h, gamma_g, hcs, zs, vs, hws, rbars, rs)
    bad_rb = list(rbars)
    bad_rb[1] = (bad_rb[1] + 1) % P.R
    assert not P.kzg_batch_check_hiding(tau_h, gamma_g, hcs, zs, vs, hws, bad_rb, rs)
    assert not P.kzg_batch_check(tau_h, hcs, zs, vs, hws, rs)  # the non-hiding check refuses hiding commitments
    kzg_h = {"tau_h": kzg["tau_h"], "gamma_g": pt(gamma_g), "comms": [pt(c) for c in hcs], "points": [hx(z) for z in zs],
             "values": [hx(v) for v in vs], "random_v": [hx(x) for x in rbars], "bad_random_v": [hx(x) for x in bad_rb],
             "witnesses": [pt(w) for w in hws], "randomizers": [hx(r) for r in rs]}
    json.dump(kzg_h, open(os.path.join(OUT, "kzg_hiding.json"), "w"), indent=0)
    # --- compressed G1 points (pyref.g1_compress / g1_decompress): points of the subgroup with both signs of y, the
    # point at infinity, an x whose x^3 + 1 is a non-residue, an x >= q, a curve point outside the order-r subgroup,
    # a malformed infinity
    comp = []
    for k in (1, 2, 3, 7, 1 << 60, rng.below(P.R, 9), rng.below(P.R, 10), P.R - 1):
        pnt = P.g1_mul(P.G1_GEN, k)
        for q in (pnt, P.g1_neg(pnt)):
            c = P.g1_compress(q)
            ok, back = P.g1_decompress(c)
            assert ok and back == q
            comp.append({"bytes": c.hex(), "ok": True, "ok_unchecked": True, "point": pt(q)})
    comp.append({"bytes": P.g1_compress(None).hex(), "ok": True, "ok_unchecked": True, "point": pt(None)})
    x = 5
    while P.fq_sqrt(x * x * x + 1) is not None:
        x += 1
    comp.append({"bytes": x.to_bytes(48, "little").hex(), "ok": False, "ok_unchecked": False, "point": pt(None)})
    comp.append({"bytes": P.Q.to_bytes(48, "little").hex(), "ok": False, "ok_unchecked": False, "point": pt(None)})
    x = 1
    while True:  # a curve point that is not in the order-r subgroup (the cofactor is large: almost every point)
        y = P.fq_sqrt(x * x * x + 1)
        if y is not None and P._g1_mul_any((x, y), P.R) is not None:
            break
        x += 1
    ok_u, pu = P.g1_decompress(P.g1_compress((x, y)), check_subgroup=False)
    assert ok_u and pu == (x, y) and not P.g1_decompress(P.g1_compress((x, y)))[0]
    comp.append({"bytes": P.g1_compress((x, y)).hex(), "ok": False, "ok_unchecked": True, "point": pt((x, y))})
    bad_inf = bytearray(P.g1_compress(None))
    bad_inf[3] = 1
    comp.append({"bytes": bytes(bad_inf).hex(), "ok": False, "ok_unchecked": False, "point": pt(None)})
    # y = 0 (x^3 = -1, here x = q - 1): the point of order two.  Its only encoding has the sign flag clear; with the flag
    # set it is non-canonical (the flag has nothing to select) and rejected even without the subgroup check
    x2 = P.Q - 1
    assert (x2 ** 3 + 1) % P.Q == 0
    comp.append({"bytes": x2.to_bytes(48, "little").hex(), "ok": False, "ok_unchecked": True, "point": pt((x2, 0))})
    flagged = bytearray(x2.to_bytes(48, "little"))
    flagged[47] |= 0x80
    assert P.g1_decompress(bytes(flagged), check_subgroup=False) == (False, None)
    comp.append({"bytes": bytes(flagged).hex(), "ok": False, "ok_unchecked": False, "point": pt(None)})
    roots = [{"a": hx(a), "root": hx(P.fq_sqrt(a)) if P.fq_sqrt(a) is not None else None}
             for a in [0, 1, 4, 15, P.Q - 1] + [rng.below(P.Q, 11 + i) for i in range(6)]]
    json.dump({"points": comp, "sqrt": roots}, open(os.path.join(OUT, "g1_compressed.json"), "w"), indent=0)
    # --- Schnorr over Edwards-BLS12: the derived generator, Poseidon-over-Fr permutations of widths 5
    # and 9, hash-to-scalar, and signatures made by pyref's signer -- valid ones and each way of spoiling one
    g = P.ed_generator()
    perms = []
    for w in (5, 9):
        st = [rng.below(P.R, 5) for _ in range(w)]
        perms.append({"width": w, "in": [hx(x) for x in st], "out": [hx(x) for x in P.ed_poseidon_permute(st)]})
    h2s = []
    for rate, m in ((4, 2), (8, 5), (8, 8), (4, 0)):
        ins = [rng.below(P.R, 5) for _ in range(m)]
        h2s.append({"rate": rate, "inputs": [hx(x) for x in ins], "scalar": hx(P.ed_hash_to_scalar(ins, rate))})
    sigs = []
    for k in range(4):
        sk, rs, nonce, msg = rng.below(P.ED_S, 5), rng.below(P.ED_S, 5), rng.below(P.ED_S, 5), rng.below(P.R, 5)
        sig, addr = P.schnorr_sign(sk, rs, nonce, [msg], g)
        assert P.schnorr_verify(addr, [msg], sig, g)
        rec = {"address": pt(addr), "message": hx(msg), "challenge": hx(sig["challenge"]), "response": hx(sig["response"]),
               "pk_sig": pt(sig["pk_sig"]), "pr_sig": pt(sig["pr_sig"]), "valid": True}
        sigs.append(rec)
        spoil = dict(rec)
        if k == 0:
            spoil["response"] = hx((sig["response"] + 1) % P.ED_S)
        elif k == 1:
            spoil["message"] = hx((msg + 1) % P.R)
        elif k == 2:
            spoil["pr_sig"] = pt(g)  # another point of the curve: the address identity and the challenge both break
        else:
            spoil["challenge"] = hx(sig["challenge"] ^ 2)
        spoil["valid"] = False
        bad_sig = {"challenge": int(spoil["challenge"], 16), "response": int(spoil["response"], 16),
                   "pk_sig": tuple(int(c, 16) for c in spoil["pk_sig"]), "pr_sig": tuple(int(c, 16) for c in spoil["pr_sig"])}
        assert not P.schnorr_verify(addr, [int(spoil["message"], 16)], bad_sig, g)
        sigs.append(spoil)
    json.dump({"curve": {"a": hx(P.ED_A), "d": hx(P.ED_D), "order": hx(P.ED_S)}, "generator": pt(g),
               "mul": [[hx(k), pt(P.ed_mul(g, k))] for k in (1, 2, 3, P.ED_S - 1, rng.below(P.ED_S, 5))],
               "poseidon_fr": perms, "hash_to_scalar": h2s, "signatures": sigs}, open(os.path.join(OUT, "schnorr.json"), "w"), indent=0)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
