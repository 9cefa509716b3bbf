"""One large MSM a few times (for rocprofv3 kernel traces): python tools/msm_one.py LOG_N"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snarkos_amd import algorithms as A, lib as L
L.init(0)
n = 1 << int(sys.argv[1])
pts = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
L.check(L.lib().mi_dbg_gen_points_device(C.c_uint64(7), C.c_size_t(n), L.dptr(pts), L.stream_ptr()), "gen")
g = torch.Generator(device="cpu").manual_seed(1)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from uniform_scalars import uniform_scalars  # uniform in [0, r)
sc = uniform_scalars(n, 1).cuda()
torch.cuda.synchronize()
for _ in range(4):
    A.VariableBase.msm(pts, sc)
torch.cuda.synchronize()
