"""Host-side spans of one batch on its own (MI_SNARK_HOSTPROF report on stderr at exit): where the time between the
last MSM kernel and the pairing launch goes.  python tools/hostprof_one.py [repetitions]"""
import os, sys, time
os.environ["MI_SNARK_HOSTPROF"] = "1"
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snarkos_amd import algorithms as A, lib as L
L.init(0)
d = np.load(os.path.join(ROOT, "tests", "golden", "vs_batch_1024.npz"))
v = A.VarunaShaped(d["vk"], d["circuits"])
p, r = torch.from_numpy(d["proofs"]).cuda(), torch.from_numpy(d["randomizers"]).cuda()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for _ in range(5):
    assert v.verify_batch(p, r)[0]
t0 = time.perf_counter()
for _ in range(reps):
    assert v.verify_batch(p, r)[0]
print("ms per batch: %.4f" % ((time.perf_counter() - t0) / reps * 1e3))
v.close()
L.lib().mi_shutdown()
