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


def one(k, fold, depth, lanes):
    sid, ok = C.c_uint64(0), C.c_uint8(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    L.check(lib.mi_vs_stream_open(C.c_uint64(v.ctx), fold, depth, lanes, C.cast(None, L.VS_EXCHANGE_FN), None, 1, C.byref(sid)), "open")
    t_open = time.perf_counter() - t0
    subs, outs = [], []
    for _ in range(k):
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
for pair_at in ("2", "4"):
    os.environ["MI_SNARK_STREAM_PAIR_AT"] = pair_at  # read when a stream is opened
    for fold, depth, lanes in ((3, 10, 10), (5, 10, 10), (6, 10, 10), (8, 10, 10)):
        best = None
        for rep in range(9):
            r = one(K, fold, depth, lanes)
            if best is None or r[4] < best[4]:
                best = r
        t_open, subs, outs, t_last, t_all = best
        print("pair_at=%s fold=%d products=%d lanes=%d  K=%d: total %.3f ms (%.3f ms/step)  open %.3f  last submit %.3f  first verdict %.3f  last verdict %.3f  close %.3f"
              % (pair_at, fold, depth, lanes, K, 1e3 * t_all, 1e3 * t_all / K, 1e3 * t_open, 1e3 * subs[-1], 1e3 * outs[0], 1e3 * t_last, 1e3 * (t_all - t_last)))
        print("   submits:", " ".join("%.2f" % (1e3 * s) for s in subs))
        print("   verdicts:", " ".join("%.2f" % (1e3 * s) for s in outs))
