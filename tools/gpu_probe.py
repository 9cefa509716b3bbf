"""Ad-hoc timing probe for the GPU box (not part of the product or the tests)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snarkos_amd import algorithms as A, lib as L

L.init(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(1)
r = C.c_double(0)
for nt in (256 * 256, 256 * 1024):
    L.check(L.lib().mi_dbg_fq_mul_throughput(C.c_size_t(nt), C.c_uint32(200), C.byref(r)), "tp")
    print("fq mul throughput threads=%d: %.2f Gmul/s" % (nt, r.value), flush=True)
names = ["msm_total", "msm_accumulate", "msm_digits", "msm_scan", "msm_scatter", "msm_chunks", "msm_combine", "msm_segments", "msm_tree", "msm_bitsums"]
for lg in [int(x) for x in os.environ.get("PROBE_LGS", "10,12,13,14,16,18,20").split(",")]:
    n = 1 << lg
    pts = torch.empty(n * 96, dtype=torch.uint8, device=dev)
    L.check(L.lib().mi_dbg_gen_points_device(C.c_uint64(7), C.c_size_t(n), L.dptr(pts), L.stream_ptr()), "gen")
    sc = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, generator=g)
    sc[:, 3] &= (1 << 60) - 1  # uniform below 2^252 < r
    sc = sc.to(dev)
    torch.cuda.synchronize()
    A.VariableBase.msm(pts, sc)
    L.prof_enable(2)
    for k in names:
        L.prof_read(k)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        A.VariableBase.msm(pts, sc)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    parts = []
    for k in names:
        ms, cnt, _ = L.prof_read(k)
        if cnt:
            parts.append("%s=%.3f" % (k.replace("msm_", ""), ms / cnt))
    L.prof_enable(False)
    print("MSM 2^%d: %.3f ms wall (%.1f Mpts/s)  kernels[ms]: %s" % (lg, dt * 1e3, n / dt / 1e6, " ".join(parts)), flush=True)
for lg in (12, 16, 20, 22):
    n = 1 << lg
    a = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, generator=g)
    a[:, 3] &= (1 << 58) - 1
    a = a.to(dev)
    d = A.EvaluationDomain(n)
    d.fft(a); d.ifft(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        d.fft(a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("NTT 2^%d fwd: %.3f ms (%.1f Melem/s, %.1f GB/s algorithmic)" % (lg, dt * 1e3, n / dt / 1e6, 64 * n / dt / 1e9), flush=True)
