"""The 1024-proof batch in the WIRE form (MI_VS_WIRE: 48-byte compressed commitments, canonical Fr), decoded inside
the lane pass: one batch on its own (latency), the decode launches' HIP-event time, and the streamed rate.
    python tools/wire_one.py [batches]          (also the workload of the rocprofv3 kernel trace in profiles/)"""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snarkos_amd import algorithms as A, lib as L


def measure(dev, nb=240, transcript="sha256"):
    d = np.load(os.path.join(ROOT, "tests", "golden", "vs_batch_1024.npz" if transcript == "sha256" else "vs_batch_1024_poseidon.npz"))
    v = A.VarunaShaped(d["vk"], d["circuits"], transcript=transcript, wire=True)
    wire = v.wire_encode(d["proofs"])
    p, r = torch.from_numpy(wire).to(dev), torch.from_numpy(d["randomizers"]).to(dev)
    ok, verdicts = v.verify_batch(p, r)
    assert ok and verdicts.all()
    L.prof_enable(2)
    L.prof_read("vs_wire_decode")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ok, _ = v.verify_batch(p, r)
    torch.cuda.synchronize()
    lat = (time.perf_counter() - t0) / 5
    ms, launches, units = L.prof_read("vs_wire_decode")
    L.prof_enable(0)
    list(v.verify_stream(((p, r) for _ in range(12))))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    oks = list(v.verify_stream(((p, r) for _ in range(nb))))
    torch.cuda.synchronize()
    dts = time.perf_counter() - t0
    assert all(oks)
    bad = wire.copy()
    bad[17 * 1728 + 16 + 512 + 48 * 3 + 5] ^= 0x20  # one damaged commitment: the batch fails, the fallback names proof 17
    ok, verdicts = v.verify_batch(torch.from_numpy(bad).to(dev), r)
    assert not ok and verdicts.sum() == 1023 and verdicts[17] == 0
    out = {"record_bytes": 1728, "points_decoded_per_proof": 11, "transcript": transcript,
           "latency_ms_one_batch_unpipelined": 1e3 * lat, "decode_ms": ms / max(1, launches),
           "streamed_proofs_per_s": 1024 * nb / dts, "streamed_ms_per_batch": 1e3 * dts / nb, "batches": nb,
           "note": "the decode (square root in Fq, sign, exact subgroup test per point; Fr canonical -> Montgomery) runs inside "
                   "every batch's lane pass, in the timed stream"}
    v.close()
    return out


if __name__ == "__main__":
    L.init(0)
    print(json.dumps(measure(torch.device("cuda", 0), int(sys.argv[1]) if len(sys.argv) > 1 else 240)))
