"""6 700 certificate signatures (100 certificates x 67 signers) a few times, for rocprofv3 kernel traces: python tools/schnorr_one.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import orc_py as O
from snarkos_amd import algorithms as A, lib as L
L.init(0)
sb = O.SigBatch(0xA1E00005, 6700, 100, 0)
d = torch.from_numpy(sb.records).cuda()
for _ in range(6):
    out = A.SchnorrBatch.verify(d)
torch.cuda.synchronize()
assert bool(out.all().item())
