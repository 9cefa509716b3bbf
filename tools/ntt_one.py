"""One large NTT round trip a few times (for rocprofv3 kernel traces): python tools/ntt_one.py LOG_N"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snarkos_amd import algorithms as A, lib as L
L.init(0)
log_n = int(sys.argv[1])
g = torch.Generator(device="cpu").manual_seed(3)
a = torch.randint(0, 1 << 62, (1 << log_n, 4), dtype=torch.int64, generator=g)
a[:, 3] &= (1 << 58) - 1
a = a.cuda()
d = A.EvaluationDomain(1 << log_n)
torch.cuda.synchronize()
for _ in range(4):
    d.fft(a)
    d.ifft(a)
torch.cuda.synchronize()
