"""-m gpu: lane passes whose batches are folded into ONE aggregate (mi_vs_accumulate_begin_merged) -- the partial of the
merged pass is, byte for byte, the partial of the concatenated batch (and so the oracle's); the native stream driver
merges under load when it has no exchange, and a failing merged product is resolved batch by batch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(mi, b, **kw):
    return mi.VarunaShaped(b.vk, b.circuits, **kw)


def _split(b, cuts, rand_bytes=32, proof_bytes=2256):
    import torch
    out, lo = [], 0
    for hi in list(cuts) + [len(b.proofs) // proof_bytes]:
        out.append((torch.from_numpy(b.proofs[lo * proof_bytes:hi * proof_bytes].copy()).cuda(),
                    torch.from_numpy(b.randomizers[lo * rand_bytes:hi * rand_bytes].copy()).cuda()))
        lo = hi
    return out


@pytest.mark.parametrize("transcript", ["sha256", "poseidon"])
def test_merged_pass_partial_is_the_concatenated_batch(orc, mi, transcript):
    import torch
    from snarkos_amd.lib import MiError
    b = orc.VSBatch(0x3E26ED, 97, 3, 0, transcript=orc.FS_POSEIDON if transcript == "poseidon" else orc.FS_SHA256)
    v = _ctx(mi, b, transcript=transcript)
    try:
        whole = v.accumulate(torch.from_numpy(b.proofs).cuda(), torch.from_numpy(b.randomizers).cuda())
        assert (whole == orc.vs_partial_from_oracle(b)).all()
        for cuts in ((40,), (1,), (96,), (33, 64), (10, 11, 90), (5, 17, 30, 31, 60, 77, 90)):
            parts = _split(b, cuts)
            merged = v.accumulate_end(v.accumulate_begin_merged(parts))
            assert (merged == whole).all(), cuts
            assert v.finalize([merged])
        # an invalid proof in the second batch: the merged aggregate fails, as the concatenated batch's does
        bad = b.proofs.copy()
        bad[70 * 2256 + 16 + 512 + 864 + 32 * 3] ^= 1
        bb = orc.VSArrays(b.vk, b.circuits, bad, b.randomizers)
        bb.transcript = b.transcript
        parts = _split(bb, (40,))
        merged = v.accumulate_end(v.accumulate_begin_merged(parts))
        assert (merged == v.accumulate(torch.from_numpy(bad).cuda(), torch.from_numpy(b.randomizers).cuda())).all()
        assert not v.finalize([merged])
        assert v.verify_batch(*parts[0])[0] and not v.verify_batch(*parts[1])[0]
        # a merged handle is a single pass's handle: ending it as a multi pass is refused, and it stays usable
        h = v.accumulate_begin_merged(_split(b, (40,)))
        with pytest.raises(MiError):
            v.accumulate_end_multi(h, 2)
        assert (v.accumulate_end(h) == whole).all()
        # one batch is not a merged pass; batches that do not fit one pass together are refused
        with pytest.raises(MiError):
            v.accumulate_begin_merged(_split(b, ()))
        big = (torch.cat([torch.from_numpy(b.proofs).cuda()] * 60), torch.cat([torch.from_numpy(b.randomizers).cuda()] * 60))  # 2 x 75 k points > 2^17
        with pytest.raises(MiError):
            v.accumulate_begin_merged([big, big])
        assert (v.accumulate(torch.from_numpy(b.proofs).cuda(), torch.from_numpy(b.randomizers).cuda()) == whole).all()
    finally:
        v.close()


def test_merged_pass_v2_shape_and_wire_records(orc, mi):
    """the same identity for a shape with hiding and degree-bound classes (four MSM problems per pass) and for wire
    records (the decode kernels serve all batches of the pass)"""
    import torch
    shape = orc.vs2_shape(hiding=True, bound_class=[-1, -1, -1, -1, -1, 0, 0, 1, 1])
    b = orc.VS2Batch(0x3E27, 50, 3, 0, shape)
    v = mi.VarunaShaped(b.vk, b.circuits, shape=b.shape, key_ext=b.ext)
    try:
        pb, rb = b.record_bytes, len(b.randomizers) // b.n
        whole = v.accumulate(torch.from_numpy(b.proofs).cuda(), torch.from_numpy(b.randomizers).cuda())
        # four MSM problems per batch: two batches side by side fill a pass (eight problems), four merged ones do not
        assert v.batches_per_pass(44) == 2 and v.batches_per_merged_pass(44) == 8 and v.batches_per_merged_pass(3000) == 2
        for cuts in ((20,), (7, 30), (5, 20, 40)):
            parts = _split(b, cuts, rand_bytes=rb, proof_bytes=pb)
            assert (v.accumulate_end(v.accumulate_begin_merged(parts)) == whole).all(), cuts
    finally:
        v.close()
    w = mi.VarunaShaped(b.vk, b.circuits, shape=b.shape, key_ext=b.ext, wire=True)
    try:
        wire = w.wire_encode(b.proofs)
        wb = w.proof_bytes
        whole = w.accumulate(torch.from_numpy(wire).cuda(), torch.from_numpy(b.randomizers).cuda())
        parts = [(torch.from_numpy(wire[:20 * wb].copy()).cuda(), torch.from_numpy(b.randomizers[:20 * rb].copy()).cuda()),
                 (torch.from_numpy(wire[20 * wb:].copy()).cuda(), torch.from_numpy(b.randomizers[20 * rb:].copy()).cuda())]
        assert (w.accumulate_end(w.accumulate_begin_merged(parts)) == whole).all()
    finally:
        w.close()


def test_stream_merges_under_load_and_resolves_a_failing_aggregate(orc, mi, monkeypatch):
    """the stream driver with sharing from the first open pass on: without an exchange the passes are merged (fewer
    pairing products than batches), every verdict is right, a batch with an invalid proof inside a merged pass fails ALONE
    and its verdict vector names the proof; MI_SNARK_STREAM_MERGE=0 gives the same answers with a product per batch"""
    import torch
    from snarkos_amd import lib as L
    good = orc.VSBatch(0x3E28, 40, 3, 0)
    v = _ctx(mi, good)
    try:
        gp, gr = torch.from_numpy(good.proofs).cuda(), torch.from_numpy(good.randomizers).cuda()
        bad_np = good.proofs.copy()
        bad_np[17 * 2256 + 16 + 512 + 864 + 32 * 9] ^= 1
        bp = torch.from_numpy(bad_np).cuda()
        pattern = [k % 7 == 3 for k in range(31)]  # batches 3, 10, 17, 24 are the invalid one
        seq = [(bp if x else gp, gr) for x in pattern]
        for merge in ("1", "0"):
            monkeypatch.setenv("MI_SNARK_STREAM_PAIR_AT", "1")
            monkeypatch.setenv("MI_SNARK_STREAM_MERGE", merge)
            before = L.stats_read()
            res = list(v.verify_stream(iter(seq), verdicts=True))
            after = L.stats_read()
            assert [ok for ok, _ in res] == [not x for x in pattern]
            for (ok, vec), x in zip(res, pattern):
                if x:
                    assert vec is not None and len(vec) == 40 and vec[17] == 0 and int(vec.sum()) == 39
            products = after["pairing_products"] - before["pairing_products"]
            if merge == "1":
                assert products < len(seq) + 4 * 3  # merged passes: fewer products than batches (+ the re-checks' own)
            # all_ok only (no verdict vectors): the same answers
            assert list(v.verify_stream(iter(seq))) == [not x for x in pattern]
    finally:
        v.close()


def test_merged_pass_at_full_size(orc, mi):
    """BASELINE configs[3] at the pass the stream driver forms under load: eight 1024-proof batches (the committed fixture
    with eight different sets of randomizers) folded into one aggregate of 8192 proofs -- the partial is the oracle's for
    the concatenated batch, the product is one, and with one invalid proof in the sixth batch the aggregate fails while
    every batch on its own gives the right verdict"""
    import os
    import torch
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vs_batch_1024.npz"))
    v = mi.VarunaShaped(d["vk"], d["circuits"])
    try:
        assert v.batches_per_merged_pass(1024) == 8
        proofs = d["proofs"]
        rnds = [np.ascontiguousarray(np.roll(d["randomizers"].reshape(-1, 32), k, axis=0)).reshape(-1) for k in range(8)]
        parts = [(torch.from_numpy(proofs).cuda(), torch.from_numpy(r).cuda()) for r in rnds]
        merged = v.accumulate_end(v.accumulate_begin_merged(parts))
        whole = orc.VSArrays(d["vk"], d["circuits"], np.tile(proofs, 8), np.concatenate(rnds))
        want = np.frombuffer(np.ascontiguousarray(orc.vs_partial_from_oracle(whole, threads=8)).tobytes(), dtype=np.uint8)
        assert (merged == want).all()
        assert v.finalize([merged])
        bad = proofs.copy()
        bad[777 * 2256 + 16 + 512 + 864 + 32 * 4] ^= 2
        parts[5] = (torch.from_numpy(bad).cuda(), parts[5][1])
        merged = v.accumulate_end(v.accumulate_begin_merged(parts))
        assert not v.finalize([merged])
        res = list(v.verify_stream(iter(parts), verdicts=True))
        assert [ok for ok, _ in res] == [k != 5 for k in range(8)]
        assert res[5][1] is not None and res[5][1][777] == 0 and int(res[5][1].sum()) == 1023
    finally:
        v.close()
