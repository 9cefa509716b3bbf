"""-m gpu: VarunaShaped-v2 (query-set table, hiding gamma_g / random_v, degree-bound classes: 2 + n_bounds pairs, three
randomizers per proof) through the same mi_vs_* entry points, against the CPU oracle (oracle/orc_algo.c vs2_*, itself
checked against oracle/pyref.py vs2_verify_one in tests/test_oracle_golden.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P2 = 3184


def _ctx(mi, b, transcript="sha256"):
    return mi.VarunaShaped(b.vk, b.circuits, transcript=transcript, shape=b.shape, key_ext=b.ext)


def _partial_points(whole):
    from helpers import jac_affine
    part = np.frombuffer(np.ascontiguousarray(whole).tobytes(), dtype=np.uint64)
    return np.stack([jac_affine(part[18 * k:18 * k + 18]) for k in range(4)]), int(part[72]), int(part[73])


def test_valid_batch_and_partial_bit_exact(orc, mi):
    import torch
    b = orc.VS2Batch(0x20B1, 72, 5, 0)
    v = _ctx(mi, b)
    try:
        ok, verdicts = v.verify_batch(b.proofs, b.randomizers)
        assert ok and verdicts.all() and (b.verify_cpu(8) == verdicts).all()
        proofs = torch.from_numpy(b.proofs).cuda(); rnd = torch.from_numpy(b.randomizers).cuda()
        whole = v.accumulate(proofs, rnd)
        ok_o, alg, tot = b.verify_aggregated(8)
        pts, n, fails = _partial_points(whole)
        assert ok_o and n == b.n and fails == 0
        assert (pts == tot).all()
        assert tot[2].any() and tot[3].any()  # both degree-bound classes carry weight
        assert (np.frombuffer(whole.tobytes(), dtype=np.uint8) == orc.vs_partial_from_oracle(b, 8)).all()
        assert v.finalize([whole])
        # shards: three ragged pieces sum to the same verdict
        cuts = [0, 17, 18, 72]
        parts = [v.accumulate(proofs[cuts[i] * P2:cuts[i + 1] * P2].contiguous(), rnd[cuts[i] * 48:cuts[i + 1] * 48].contiguous())
                 for i in range(3)]
        assert v.finalize(parts)
    finally:
        v.close()


def test_corrupted_batch_names_the_bad_proofs(orc, mi):
    b = orc.VS2Batch(0x20B2, 80, 4, 200)
    assert 0 < b.expected.sum() < b.n
    v = _ctx(mi, b)
    try:
        ok, verdicts = v.verify_batch(b.proofs, b.randomizers)
        assert not ok and (verdicts == b.expected).all() and (verdicts == b.verify_cpu(8)).all()
    finally:
        v.close()


@pytest.mark.parametrize("field", ["random_v", "shifted", "commitment_of_bounded", "evaluation"])
def test_each_new_proof_field_is_bound(orc, mi, field):
    """flipping one bit in random_v, a shifted commitment, a degree-bounded commitment or an evaluation turns exactly
    that proof invalid"""
    b = orc.VS2Batch(0x20B3, 24, 3, 0)
    off = {"random_v": 2256 + 5, "shifted": 2256 + 64 + 96 * 6 + 3, "commitment_of_bounded": 16 + 512 + 96 * 7 + 2,
           "evaluation": 16 + 512 + 864 + 32 * 13}[field]
    bad = b.proofs.copy()
    bad[11 * P2 + off] ^= 1
    v = _ctx(mi, b)
    try:
        ok, verdicts = v.verify_batch(bad, b.randomizers)
        want = np.ones(24, dtype=np.uint8); want[11] = 0
        cpu = orc.VS2Batch.wrap(b.vk, b.ext, b.shape, b.circuits, bad, b.randomizers).verify_cpu(8)
        assert (cpu == want).all()
        assert not ok and (verdicts == want).all()
        if field == "shifted":
            # polynomial 1 has no degree bound in the default shape: its shifted slot is validated (a point of the curve)
            # but enters no sum -- another valid point there changes nothing, an off-curve one is malformed
            same = b.proofs.copy()
            o1, o6 = 11 * P2 + 2256 + 64 + 96 * 1, 11 * P2 + 2256 + 64 + 96 * 6
            same[o1:o1 + 96] = b.proofs[o6:o6 + 96]
            ok, verdicts = v.verify_batch(same, b.randomizers)
            assert ok and verdicts.all()
            same[o1 + 3] ^= 1
            ok, verdicts = v.verify_batch(same, b.randomizers)
            cpu = orc.VS2Batch.wrap(b.vk, b.ext, b.shape, b.circuits, same, b.randomizers).verify_cpu(8)
            assert not ok and (verdicts == want).all() and (cpu == want).all()
    finally:
        v.close()


@pytest.mark.parametrize("hiding,n_bounds,bound_class,lc", [
    (False, 0, [-1] * 9, None),                              # the v1 relation in v2 records
    (True, 0, [-1] * 9, None),                               # hiding only
    (False, 1, [-1, -1, 0, -1, -1, -1, -1, -1, 0], None),    # one class, K = 3
    (True, 2, [1, -1, -1, 0, -1, 1, -1, 0, 0], None),        # K = 4, classes of unequal size
    (True, 2, [1, -1, -1, 0, -1, 1, -1, 0, 0], [0, 0, 1, 2, 3, 0, 1, 3, 2] + [3, 1, 0, 2] * 3),  # + challenge-dependent LC coefficients
    (False, 0, [-1] * 9, [0, 0] + [3] * 19),                 # every other polynomial times both vanishing values
])
def test_shapes(orc, mi, hiding, n_bounds, bound_class, lc):
    import torch
    q = [0, 0, 1, 0, 1, 1, 0, 1, 0] + [1, 0] * 6
    shape = orc.vs2_shape(q, bound_class, hiding, n_bounds, lc_kind=lc)
    b = orc.VS2Batch(0x20B4 + n_bounds, 40, 3, 128, shape=shape)
    assert 0 < b.expected.sum() < b.n
    v = _ctx(mi, b)
    try:
        ok, verdicts = v.verify_batch(b.proofs, b.randomizers)
        assert not ok and (verdicts == b.expected).all()
        good = orc.VS2Batch(0x20B9, 40, 3, 0, shape=shape)
        v2 = _ctx(mi, good)
        try:
            whole = v2.accumulate(torch.from_numpy(good.proofs).cuda(), torch.from_numpy(good.randomizers).cuda())
            assert (np.frombuffer(whole.tobytes(), dtype=np.uint8) == orc.vs_partial_from_oracle(good, 8)).all()
            assert v2.finalize([whole])
            pts, _, _ = _partial_points(whole)
            for c in range(2):
                assert pts[2 + c].any() == (c < n_bounds)
        finally:
            v2.close()
    finally:
        v.close()


def test_v1_shape_in_v2_records_equals_v1(orc, mi):
    """the v1 relation (fixed query table, no hiding, no bounds) through mi_vs_ctx_create_v2 gives the partial the v1
    context gives on the same proofs (total_c, total_w; the third randomizer is unused)"""
    import torch
    b1 = orc.VSBatch(0x20B5, 32, 3, 0)
    shape = orc.vs2_shape(None, [-1] * 9, False, 0)
    rec = np.zeros((32, P2), dtype=np.uint8)
    rec[:, :2256] = b1.proofs.reshape(32, 2256)
    rnd = np.zeros((32, 48), dtype=np.uint8)
    rnd[:, :32] = b1.randomizers.reshape(32, 32)
    rnd[:, 32:] = 7
    v1 = mi.VarunaShaped(b1.vk, b1.circuits)
    v2 = mi.VarunaShaped(b1.vk, b1.circuits, shape=shape)
    try:
        a = v1.accumulate(torch.from_numpy(b1.proofs).cuda(), torch.from_numpy(b1.randomizers).cuda())
        c = v2.accumulate(torch.from_numpy(rec.reshape(-1)).cuda(), torch.from_numpy(rnd.reshape(-1)).cuda())
        assert (np.asarray(a) == np.asarray(c)).all()
        assert v2.finalize([c])
    finally:
        v1.close(); v2.close()


def test_poseidon_transcript_with_v2(orc, mi):
    b = orc.VS2Batch(0x20B6, 36, 3, 150, transcript=orc.FS_POSEIDON)
    assert 0 < b.expected.sum() < b.n
    v = _ctx(mi, b, "poseidon")
    try:
        ok, verdicts = v.verify_batch(b.proofs, b.randomizers)
        assert not ok and (verdicts == b.expected).all()
    finally:
        v.close()
        orc.lib().orc_vs_set_transcript(orc.FS_SHA256)


def test_stream_and_block(orc, mi):
    """the pipelined stream and the in-library block path take v2 records"""
    import torch
    good = orc.VS2Batch(0x20B7, 48, 3, 0)
    v = _ctx(mi, good)
    try:
        gp = torch.from_numpy(good.proofs).cuda(); gr = torch.from_numpy(good.randomizers).cuda()
        bad_np = good.proofs.copy(); bad_np[9 * P2 + 2256 + 64 + 96 * 8] ^= 1
        bp = torch.from_numpy(bad_np).cuda()
        seq = [(gp, gr), (bp, gr), (gp, gr), (gp, gr), (bp, gr), (gp, gr)]
        want = [True, False, True, True, False, True]
        assert list(v.verify_stream(iter(seq))) == want
        ok, verdicts = v.verify_block(bad_np, good.randomizers)
        w = np.ones(48, dtype=np.uint8); w[9] = 0
        assert not ok and (verdicts == w).all()
        ok, verdicts = v.verify_block(good.proofs, good.randomizers)
        assert ok and verdicts.all()
    finally:
        v.close()


def test_full_size_batch_1024_v2(orc, mi):
    """configs[3]'s size with the v2 shape: the committed fixture (tests/golden/vs2_batch_1024.npz, oracle/gen_vs_fixture.py),
    valid then ~1 % corrupted, verdict vector against the oracle"""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vs2_batch_1024.npz"))
    v = mi.VarunaShaped(d["vk"], d["circuits"], shape=d["shape"], key_ext=d["ext"])
    try:
        ok, verdicts = v.verify_batch(d["proofs"], d["randomizers"])
        assert ok and verdicts.all()
        bad = d["proofs"].copy()
        idx = np.arange(3, 1024, 97)
        for k, i in enumerate(idx):
            bad[i * P2 + (2256 + 64 + 96 * 5 + 1, 2256 + 9, 16 + 512 + 864 + 32 * 4)[k % 3]] ^= 1
        ok, verdicts = v.verify_batch(bad, d["randomizers"])
        want = np.ones(1024, dtype=np.uint8); want[idx] = 0
        assert not ok and (verdicts == want).all()
        cpu = orc.VS2Batch.wrap(d["vk"], d["ext"], d["shape"], d["circuits"], bad, d["randomizers"]).verify_cpu(16)
        assert (cpu == want).all()
    finally:
        v.close()


@pytest.mark.parametrize("transcript", ["sha256", "poseidon"])
def test_instances_per_proof(orc, mi, transcript):
    """per-circuit batch sizes (mi_vs_proof3): proofs with 1..4 instances; partial bit-exact against the oracle, verdicts
    on a corrupted batch (base and further instances spoiled), counts out of range, the fallback and the stream"""
    import torch
    tr = orc.FS_POSEIDON if transcript == "poseidon" else orc.FS_SHA256
    P3 = 5504
    try:
        sh = orc.vs2_shape(None, None, True, 2, lc_kind=[0, 0, 1, 2, 3, 0, 1, 3, 2] + [3, 1, 0, 2] * 3, max_instances=4)
        good = orc.VS2Batch(0x31A, 48, 3, 0, shape=sh, transcript=tr)
        v = _ctx(mi, good, transcript)
        try:
            assert v.proof_bytes == P3
            gp, gr = torch.from_numpy(good.proofs).cuda(), torch.from_numpy(good.randomizers).cuda()
            whole = v.accumulate(gp, gr)
            assert (np.frombuffer(whole.tobytes(), dtype=np.uint8) == orc.vs_partial_from_oracle(good, 8)).all()
            assert v.finalize([whole])
            ok, verdicts = v.verify_batch(good.proofs, good.randomizers)
            assert ok and verdicts.all()
            bad = orc.VS2Batch(0x31B, 64, 3, 300, shape=sh, transcript=tr)
            assert 0 < bad.expected.sum() < 64
        finally:
            v.close()
        v = _ctx(mi, bad, transcript)
        try:
            ok, verdicts = v.verify_batch(bad.proofs, bad.randomizers)
            assert not ok and (verdicts == bad.expected).all() and (verdicts == bad.verify_cpu(8)).all()
            # instance counts out of range
            good2 = orc.VS2Batch(0x31B, 64, 3, 0, shape=sh, transcript=tr)   # same keys as `bad` (same seed)
            p = good2.proofs.copy()
            p[7 * P3 + 3184] = 9
            p[8 * P3 + 3184] = 0
            ok, verdicts = v.verify_batch(p, good2.randomizers)
            w = np.ones(64, dtype=np.uint8); w[7] = w[8] = 0
            cpu = orc.VS2Batch.wrap(good2.vk, good2.ext, sh, good2.circuits, p, good2.randomizers, transcript=tr).verify_cpu(8)
            assert (cpu == w).all() and not ok and (verdicts == w).all()
            dp, dr = torch.from_numpy(p).cuda(), torch.from_numpy(good2.randomizers).cuda()
            g2p = torch.from_numpy(good2.proofs).cuda()
            got = list(v.verify_stream(iter([(g2p, dr), (dp, dr)]), verdicts=True))
            assert got[0][0] and not got[1][0] and (got[1][1] == w).all()
        finally:
            v.close()
    finally:
        orc.lib().orc_vs_set_transcript(orc.FS_SHA256)


def test_v2_batches_share_a_pass(orc, mi, monkeypatch):
    """a shape with two degree-bound classes brings four MSMs per batch: two such batches share one lane pass (eight
    problems), three do not; partials byte for byte those of single passes; the stream pairs them under load"""
    import torch
    from snarkos_amd.lib import MiError
    b = orc.VS2Batch(0x20C1, 44, 3, 0)
    v = _ctx(mi, b)
    try:
        p, r = torch.from_numpy(b.proofs).cuda(), torch.from_numpy(b.randomizers).cuda()
        p2_np = b.proofs[: 30 * P2].copy(); p2_np[9 * P2 + 16 + 512 + 864 + 32 * 9] ^= 1
        p2, r2 = torch.from_numpy(p2_np).cuda(), torch.from_numpy(np.ascontiguousarray(b.randomizers[14 * 48:44 * 48])).cuda()
        s1, s2 = v.accumulate(p, r), v.accumulate(p2, r2)
        assert (np.frombuffer(s1.tobytes(), dtype=np.uint8) == orc.vs_partial_from_oracle(b, 8)).all()
        assert v.batches_per_pass(44) == 2
        for first, second, w1, w2 in ((p, p2, s1, s2), (p2, p, s2, s1)):
            rr1, rr2 = (r, r2) if first is p else (r2, r)
            got = v.accumulate_end_multi(v.accumulate_begin_multi([(first, rr1), (second, rr2)]), 2)
            assert (got[0] == w1).all() and (got[1] == w2).all()
        with pytest.raises(MiError):
            v.accumulate_begin_multi([(p, r)] * 3)
        assert v.finalize([s1]) and not v.finalize([s2])
        monkeypatch.setenv("MI_SNARK_STREAM_PAIR_AT", "1")
        seq = [(p, r), (p, r), (p2, r2), (p, r), (p, r), (p2, r2), (p, r)]
        assert list(v.verify_stream(iter(seq))) == [q is p for q, _ in seq]
    finally:
        v.close()
