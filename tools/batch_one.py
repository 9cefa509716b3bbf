"""One 1024-proof batch on its own, a few times (for rocprofv3 kernel traces of the latency path: derive, the MSM launch
sequence, the pairing product in its two-launch form), then the same batch with 1 % invalid proofs (the fallback):
python tools/batch_one.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snarkos_amd import algorithms as A, lib as L
L.init(0)
d = np.load(os.path.join(ROOT, "tests", "golden", "vs_batch_1024.npz"))
v = A.VarunaShaped(d["vk"], d["circuits"])
p, r = torch.from_numpy(d["proofs"]).cuda(), torch.from_numpy(d["randomizers"]).cuda()
bad = d["proofs"].copy()
for i in range(5, 1024, 101):
    bad[i * L.VS_PROOF_BYTES + 16 + 512 + 864 + 32 * 7] ^= 1
pb = torch.from_numpy(bad).cuda()
for _ in range(8):
    assert v.verify_batch(p, r)[0]
for _ in range(4):
    assert not v.verify_batch(pb, r)[0]
torch.cuda.synchronize()
v.close()
