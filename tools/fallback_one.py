"""A 1024-proof batch with 11 corrupted proofs, verified a few times (for rocprofv3 kernel traces of the fallback path)."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snarkos_amd import algorithms as A, lib as L
L.init(0)
d = np.load(os.path.join(ROOT, "tests", "golden", "vs_batch_1024.npz"))
bad = d["proofs"].copy()
for i in range(5, 1024, 101):
    bad[i * 2256 + 16 + 512 + 864 + 32 * 7] ^= 1
v = A.VarunaShaped(d["vk"], d["circuits"])
p = torch.from_numpy(bad).cuda(); r = torch.from_numpy(d["randomizers"]).cuda()
for _ in range(4):
    ok, verdicts = v.verify_batch(p, r)
assert not ok and verdicts.sum() == 1024 - 11
