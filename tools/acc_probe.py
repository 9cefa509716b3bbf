"""Throughput of the accumulate stage alone (no pairing products) vs. the full pipeline, at several lane depths
(ad-hoc probe: is the steady state bound by the accumulation lanes, by the pairing products, or by the host?)."""
import os, sys, time, threading, queue
os.environ.setdefault("GPU_MAX_HW_QUEUES", sys.argv[1] if len(sys.argv) > 1 else "16")
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
K = 60
for depth in (1, 4, 6, 8):
    lanes = threading.Semaphore(depth); hq = queue.Queue()
    def closer():
        while True:
            h = hq.get()
            if h is None: return
            v.accumulate_end(h); lanes.release()
    th = threading.Thread(target=closer); th.start()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(K):
        lanes.acquire(); hq.put(v.accumulate_begin(p, r))
    hq.put(None); th.join(); torch.cuda.synchronize()
    print("accumulate only, depth %d: %.3f ms/batch" % (depth, 1e3 * (time.perf_counter() - t0) / K))
for depth in (4, 6, 8):
    for fd in (4, 6, 8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        oks = list(v.verify_stream(((p, r) for _ in range(K)), None, depth=fd, acc_depth=depth))
        torch.cuda.synchronize()
        assert all(oks)
        print("full pipeline, lanes %d, products in flight %d: %.3f ms/batch" % (depth, fd, 1e3 * (time.perf_counter() - t0) / K))
