"""-m gpu: compressed G1 points (mi_g1_decompress / mi_g1_compress) against the committed vectors of
tests/golden/g1_compressed.json (pyref) and the C oracle on random batches -- bit-exact points, equal verdicts.
Convention [RECALL] (include/mi355x_snark.h), parity unpinned."""
import numpy as np
import pytest

import golden_util as G

pytestmark = pytest.mark.gpu


def test_golden_vectors(orc, mi):
    d = G.load("g1_compressed.json")
    data = np.stack([np.frombuffer(bytes.fromhex(c["bytes"]), dtype=np.uint8) for c in d["points"]])
    for check, key in ((True, "ok"), (False, "ok_unchecked")):
        pts, ok = mi.G1Codec.decompress(data, check)
        for c, p, o in zip(d["points"], pts, ok):
            assert bool(o) == c[key], c["bytes"]
            assert orc.g1_to_ints(p) == (G.pt(c["point"]) if c[key] else None)
    good = [c for c in d["points"] if c["ok"]]
    back = mi.G1Codec.compress(G.g1_arr([c["point"] for c in good]))
    assert [bytes(b).hex() for b in back] == [c["bytes"] for c in good]


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 11264])
def test_random_batches_against_the_oracle(orc, mi, n):
    """n points k_i G (the fixed-base batch of the oracle), both signs, some encodings damaged: a bit of x flipped (mostly
    a non-residue or a point off the subgroup), the sign flag flipped (the other root: still a valid point), the
    infinity flag set on a non-zero x.  11 264 = the commitments of a 1024-proof batch."""
    import torch
    rng = np.random.default_rng(n)
    k = rng.integers(1, 2**63, size=(n, 4), dtype=np.uint64)
    k[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
    pts = np.zeros((n, 12), dtype=np.uint64)
    kk = np.ascontiguousarray(k)
    fr = np.zeros((n, 4), dtype=np.uint64)
    for i in range(n if n <= 70 else 70):  # Montgomery scalars for the oracle's fixed-base batch
        fr[i] = orc.fr_from_ints([orc.limbs_to_int(kk[i])])[0]
    if n > 70:
        fr[70:] = fr[np.arange(n - 70) % 70]  # repeat: the arithmetic does not care, the oracle call stays short
    orc.lib().orc_g1_fixed_base_batch(orc._p(pts), orc._p(fr), n)
    comp = orc.g1_compress(pts)
    damaged = comp.copy()
    for i in range(0, n, 3):
        kind = (i // 3) % 3
        if kind == 0:
            damaged[i, int(rng.integers(0, 46))] ^= np.uint8(1 << int(rng.integers(0, 8)))
        elif kind == 1:
            damaged[i, 47] ^= 0x80
        else:
            damaged[i, 47] |= 0x40
    for data in (comp, damaged):
        want_p, want_ok = orc.g1_decompress(data, True) if n <= 1000 else (None, None)
        got_p, got_ok = mi.G1Codec.decompress(data, True)
        if want_p is not None:
            assert (got_ok == want_ok).all() and (got_p == want_p).all()
        else:  # the oracle's subgroup check is slow: sample 300 encodings of the big batch
            idx = rng.choice(n, 300, replace=False)
            wp, wo = orc.g1_decompress(data[idx], True)
            assert (got_ok[idx] == wo).all() and (got_p[idx] == wp).all()
        if data is comp:
            assert got_ok.all() and (got_p == pts).all() and (mi.G1Codec.compress(got_p) == comp).all()
        # device-resident form, without the subgroup check: the verdict can only turn to 1 for off-subgroup points
        d_data = torch.from_numpy(np.ascontiguousarray(data)).cuda()
        d_p, d_ok = mi.G1Codec.decompress(d_data, False)
        d_ok = d_ok.cpu().numpy()
        assert (d_ok >= got_ok).all()
        same = got_ok == 1
        assert (d_p.cpu().numpy().view(np.uint64)[same] == got_p[same]).all()
