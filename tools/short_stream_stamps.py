"""Verdict arrival times of a short stream (the driver's 20 steps): when each batch was submitted and when its verdict came out."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snarkos_amd import algorithms as A, lib as L
L.init(0)
d = np.load(os.path.join(ROOT, "tests", "golden", "vs_batch_1024.npz"))
v = A.VarunaShaped(d["vk"], d["circuits"])
p, r = torch.from_numpy(d["proofs"]).cuda(), torch.from_numpy(d["randomizers"]).cuda()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
list(v.verify_stream(((p, r) for _ in range(5))))
for rep in range(3):
    torch.cuda.synchronize()
    sub, out = [], []
    t0 = time.perf_counter()
    def gen():
        for _ in range(K):
            sub.append(time.perf_counter() - t0)
            yield p, r
    for ok in v.verify_stream(gen()):
        out.append(time.perf_counter() - t0)
    print("submitted at (ms):", " ".join("%.2f" % (1e3 * x) for x in sub))
    print("verdicts at  (ms):", " ".join("%.2f" % (1e3 * x) for x in out))
v.close()
