"""MSM wall time around the small/large path crossover (ad-hoc probe; MI_SNARK_MSM_SMALL_LOG picks the limit)."""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snarkos_amd import algorithms as A, lib as L
L.init(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(1)
for lg in [int(x) for x in os.environ.get("PROBE_LGS", "13,14,15,16,17").split(",")]:
    n = 1 << lg
    pts = torch.empty(n * 96, dtype=torch.uint8, device=dev)
    L.check(L.lib().mi_dbg_gen_points_device(C.c_uint64(7), C.c_size_t(n), L.dptr(pts), L.stream_ptr()), "gen")
    sc = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, generator=g); sc[:, 3] &= (1 << 58) - 1; sc = sc.to(dev)
    torch.cuda.synchronize()
    A.VariableBase.msm(pts, sc); A.VariableBase.msm(pts, sc)
    t0 = time.perf_counter()
    for _ in range(20): A.VariableBase.msm(pts, sc)
    torch.cuda.synchronize()
    print("limit 2^%s  MSM 2^%d: %.3f ms" % (os.environ.get("MI_SNARK_MSM_SMALL_LOG", "16"), lg, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
