"""One batch on its own (synchronous verify_batch): where do the milliseconds go?  (ad-hoc probe)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snarkos_amd import algorithms as A, lib as L
L.init(0)
d = np.load(os.path.join(ROOT, "tests", "golden", "vs_batch_1024.npz"))
v = A.VarunaShaped(d["vk"], d["circuits"])
p = torch.from_numpy(d["proofs"]).cuda(); r = torch.from_numpy(d["randomizers"]).cuda()
for _ in range(5):
    assert v.verify_batch(p, r)[0]
def timed(f, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
print("verify_batch      %.3f ms" % timed(lambda: v.verify_batch(p, r)))
print("accumulate        %.3f ms" % timed(lambda: v.accumulate(p, r)))
part = v.accumulate(p, r)
print("finalize          %.3f ms" % timed(lambda: v.finalize([part])))
print("finalize_async+wait %.3f ms" % timed(lambda: v.finalize_wait(v.finalize_async([part]))))
L.prof_enable(2)
for _ in range(5): v.verify_batch(p, r)
for k in ("vs_derive", "msm_total", "msm_digits", "msm_scan", "msm_scatter", "msm_accumulate", "msm_bitsums", "pairing"):
    ms, launches, units = L.prof_read(k)
    if launches: print("  %-16s %.3f ms x %d" % (k, ms / launches, launches))
