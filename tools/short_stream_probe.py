"""Timeline of a SHORT stream (the driver's `bench.py --steps 20`): when each submit returns and when each verdict
arrives, relative to the start of the timed region, for a few pipeline shapes.  Usage: python tools/short_stream_probe.py [K]"""
import ctypes as C
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snarkos_amd import algorithms as A  # noqa: E402
from snarkos_amd import lib as L  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
d = np.load(os.path.join(ROOT, "tests", "golden", "vs_batch_1024.npz"))
dev = torch.device("cuda", 0)
L.init(0)
v = A.VarunaShaped(d["vk"], d["circuits"])
pp, pr = torch.from_numpy(d["proofs"]).to(dev), torch.from_numpy(d["randomizers"]).to(dev)
lib = L.lib()


def one(k, fold, depth, lanes, gap=0.0):
    sid, ok = C.c_uint64(0), C.c_uint8(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    L.check(lib.mi_vs_stream_open(C.c_uint64(v.ctx), fold, depth, lanes, C.cast(None, L.VS_EXCHANGE_FN), None, 1, C.byref(sid)), "open")
    t_open = time.perf_counter() - t0
    subs, outs = [], []
    for j in range(k):
        if gap and j:
            t_next = t0 + t_open + j * gap
            while time.perf_counter() < t_next:
                pass
        L.check(lib.mi_vs_stream_submit(sid, L.dptr(pp), C.c_size_t(1024), L.dptr(pr), L.stream_ptr(None)), "submit")
        subs.append(time.perf_counter() - t0)
        while lib.mi_vs_stream_next(sid, 0, C.byref(ok)) == L.MI_OK:
            outs.append(time.perf_counter() - t0)
    L.check(lib.mi_vs_stream_finish(sid), "finish")
    while lib.mi_vs_stream_next(sid, 1, C.byref(ok)) == L.MI_OK:
        assert ok.value
        outs.append(time.perf_counter() - t0)
    t_last = time.perf_counter() - t0
    lib.mi_vs_stream_close(sid)
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    return t_open, subs, outs, t_last, t_all


list(v.verify_stream(((pp, pr) for _ in range(5))))
import statistics
cfgs = [tuple(int(x) for x in c.split(",")) for c in (sys.argv[2].split() if len(sys.argv) > 2 else ["10,10"])]
for cfg in cfgs:
    lanes, depth = cfg[0], cfg[1]
    fold = cfg[2] if len(cfg) > 2 else 3
    runs = [one(K, fold, depth, lanes) for rep in range(60)]
    tot = sorted(1e3 * r[4] for r in runs)
    print("lanes=%d products=%d fold=%d K=%d: median %.2f ms  p90 %.2f  max %.2f  runs over 1.3 x median: %d of %d"
          % (lanes, depth, fold, K, statistics.median(tot), tot[int(0.9 * len(tot))], tot[-1], sum(t > 1.3 * statistics.median(tot) for t in tot), len(tot)))
    med = sorted(runs, key=lambda r: r[4])[len(runs) // 2]
    print("   median run: open %.2f  close %.2f" % (1e3 * med[0], 1e3 * (med[4] - med[3])))
    print("   submits:", " ".join("%.2f" % (1e3 * s) for s in med[1]))
    print("   verdicts:", " ".join("%.2f" % (1e3 * s) for s in med[2]))
