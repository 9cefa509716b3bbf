"""What folding the two batches of a lane pass into ONE aggregate would buy: the streamed rate of 2048-proof batches, one
per pass, against 1024-proof batches two per pass (same kernels, same points per pass; the merged form folds and pairs
once per pass).  python tools/merged_probe.py [batches]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snarkos_amd import algorithms as A, lib as L
L.init(0)
d = np.load(os.path.join(ROOT, "tests", "golden", "vs_batch_1024.npz"))
v = A.VarunaShaped(d["vk"], d["circuits"])
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
p1, r1 = torch.from_numpy(d["proofs"]).cuda(), torch.from_numpy(d["randomizers"]).cuda()
p2, r2 = torch.cat([p1, p1]), torch.cat([r1, r1])


def run(p, r, count, **kw):
    for _ in v.verify_stream(((p, r) for _ in range(20)), **kw):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ok = list(v.verify_stream(((p, r) for _ in range(count)), **kw))
    dt = time.perf_counter() - t0
    assert all(ok)
    return dt


for rep in range(2):
    for n in (400, 20):
        a = run(p1, r1, n)
        print("1024-proof batches x %d: %.3e proofs/s" % (n, 1024 * n / a))
        os.environ["MI_SNARK_STREAM_PAIR_AT"] = "0"
        b = run(p2, r2, n // 2)
        del os.environ["MI_SNARK_STREAM_PAIR_AT"]
        print("2048-proof batches x %d (one per pass): %.3e proofs/s" % (n // 2, 2048 * (n // 2) / b))
v.close()
