"""Where does the host thread spend its time per batch in the pipelined loop?  (ad-hoc probe)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snarkos_amd import algorithms as A, lib as L
L.init(0)
d = np.load(os.path.join(ROOT, "tests", "golden", "vs_batch_1024.npz"))
v = A.VarunaShaped(d["vk"], d["circuits"])
p = torch.from_numpy(d["proofs"]).cuda(); r = torch.from_numpy(d["randomizers"]).cuda()
for _ in range(3):
    assert v.verify_batch(p, r)[0]
T = {"begin": 0.0, "end": 0.0, "fin_async": 0.0, "wait": 0.0}
K = 40
pending = []; prev = None
torch.cuda.synchronize(); t_all = time.perf_counter()
for k in range(K):
    t = time.perf_counter(); cur = v.accumulate_begin(p, r); T["begin"] += time.perf_counter() - t
    if prev is not None:
        t = time.perf_counter(); part = v.accumulate_end(prev); T["end"] += time.perf_counter() - t
        t = time.perf_counter(); pending.append(v.finalize_async([part])); T["fin_async"] += time.perf_counter() - t
        if len(pending) >= 4:
            t = time.perf_counter(); assert v.finalize_wait(pending.pop(0)); T["wait"] += time.perf_counter() - t
    prev = cur
part = v.accumulate_end(prev); pending.append(v.finalize_async([part]))
while pending:
    assert v.finalize_wait(pending.pop(0))
torch.cuda.synchronize(); tot = time.perf_counter() - t_all
print("per batch ms: total %.3f" % (1e3 * tot / K), {k: round(1e3 * x / K, 3) for k, x in T.items()})
