"""The small-MSM path (per-bit sums, host fold) against the large one around its limit: mi_msm_g1_device at 2^15 .. 2^18
with MI_SNARK_MSM_SMALL_LOG = 16 (default) and 17 / 18, each in a child process.  python tools/msm_small_limit.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import time, numpy as np, torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from snarkos_amd import algorithms as A, lib as L
    from uniform_scalars import uniform_scalars
    L.init(0)
    import ctypes as C
    for log_n in (15, 16, 17, 18):
        n = 1 << log_n
        pts = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
        L.check(L.lib().mi_dbg_gen_points_device(C.c_uint64(7), C.c_size_t(n), L.dptr(pts), L.stream_ptr()), "gen")
        sc = uniform_scalars(n, 5).cuda()
        r0 = A.VariableBase.msm(pts, sc)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        R = 10
        for _ in range(R):
            A.VariableBase.msm(pts, sc)
        torch.cuda.synchronize()
        print("2^%d: %.3f ms  result %s" % (log_n, (time.perf_counter() - t0) / R * 1e3, r0.tobytes()[:8].hex()), flush=True)
    sys.exit(0)
for lim in ("16", "17", "18"):
    e = dict(os.environ); e["MI_SNARK_MSM_SMALL_LOG"] = lim
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    print("small path up to 2^%s:" % lim)
    print(out.stdout.strip() or out.stderr.strip()[-600:], flush=True)
