"""-m gpu: batch-certificate Schnorr signatures over Edwards-BLS12 (reference
node/bft/src/helpers/storage.rs:465, node/bft/src/gateway.rs:1391, account/src/lib.rs:76-88) through the C ABI against the
CPU oracle (oracle/orc_schnorr.c, itself checked against oracle/pyref.py in tests/test_oracle_golden.py).  The scheme is
[RECALL], generator and Poseidon constants are this repo's own: parity unpinned."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RB = 288


def test_valid_and_spoilt_signatures_verdict_for_verdict(orc, mi):
    good = orc.SigBatch(0x51A0, 200, 7, 0)
    v = mi.SchnorrBatch.verify(good.records)
    assert v.all() and (good.verify_cpu(8) == v).all()
    bad = orc.SigBatch(0x51A1, 333, 9, 250)   # a ragged count, a quarter of the records spoilt in every way the generator knows
    assert 0 < bad.expected.sum() < bad.n
    v = mi.SchnorrBatch.verify(bad.records)
    assert (v == bad.expected).all() and (v == bad.verify_cpu(8)).all()


def test_every_field_is_bound_and_malformed_records_are_rejected_alike(orc, mi):
    import pyref as P
    b = orc.SigBatch(0x51A2, 64, 4, 0)
    recs = b.records.copy().reshape(64, RB)
    # one bit in each field of a different record
    offs = {"address.x": 3, "address.y": 32 + 5, "message": 64 + 1, "challenge": 96 + 2, "response": 128 + 7, "pk_sig.x": 160 + 9,
            "pk_sig.y": 192 + 4, "pr_sig.x": 224 + 6, "pr_sig.y": 256 + 8}
    want = np.ones(64, dtype=np.uint8)
    for k, (name, off) in enumerate(offs.items()):
        recs[3 + 5 * k, off] ^= 1
        want[3 + 5 * k] = 0
    # malformed: a scalar >= s, a coordinate >= r, a message >= r
    order = np.frombuffer(P.ED_S.to_bytes(32, "little"), dtype=np.uint8)
    recs[50, 128:160] = order                       # response == s
    recs[52, 0:32] = 0xFF                           # address.x >= r
    recs[54, 64:96] = 0xFF                          # message >= r
    want[[50, 52, 54]] = 0
    flat = recs.reshape(-1)
    cpu = b.verify_cpu(8, flat)
    assert (cpu == want).all()
    assert (mi.SchnorrBatch.verify(flat) == want).all()


def test_device_entry_point_and_a_subdag_sized_batch(orc, mi):
    import torch
    b = orc.SigBatch(0x51A3, 4096, 100, 10)
    d = torch.from_numpy(b.records).cuda()
    out = mi.SchnorrBatch.verify(d)
    torch.cuda.synchronize()
    assert (out.cpu().numpy() == b.expected).all()
    assert mi.SchnorrBatch.verify(b.records[:RB].copy())[0] == b.expected[0]   # a batch of one


def test_golden_signatures_from_pyref(orc, mi):
    """tests/golden/schnorr.json: signatures made and judged by the Python big-integer restatement, through the C ABI"""
    import golden_util as G
    from test_oracle_golden import _sig_records_from_golden
    d = G.load("schnorr.json")
    recs = _sig_records_from_golden(orc, d["signatures"])
    assert list(mi.SchnorrBatch.verify(recs)) == [int(s["valid"]) for s in d["signatures"]]
