"""Context create / stream / destroy cycles: device memory and stream leaks (ad-hoc probe, run on the GPU box)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snarkos_amd import algorithms as A, lib as L
L.init(0)
d = np.load(os.path.join(ROOT, "tests", "golden", "vs_batch_1024.npz"))
p = torch.from_numpy(d["proofs"]).cuda(); r = torch.from_numpy(d["randomizers"]).cuda()
def cycle():
    v = A.VarunaShaped(d["vk"], d["circuits"])
    assert all(v.verify_stream(((p, r) for _ in range(30))))
    assert v.verify_batch(p, r)[0]
    v.close()
cycle(); torch.cuda.synchronize()
free0, total = torch.cuda.mem_get_info()
rates = []
for k in range(40):
    t0 = time.perf_counter(); cycle(); rates.append(time.perf_counter() - t0)
torch.cuda.synchronize()
free1, _ = torch.cuda.mem_get_info()
print("40 cycles: device memory delta %.1f MiB, cycle time first %.1f ms / last %.1f ms" % ((free0 - free1) / 2**20, rates[0] * 1e3, rates[-1] * 1e3))
v = A.VarunaShaped(d["vk"], d["circuits"])
t0 = time.perf_counter(); n = 400
assert all(v.verify_stream(((p, r) for _ in range(n))))
print("throughput after the cycles: %.0f proofs/s" % (1024 * n / (time.perf_counter() - t0)))
