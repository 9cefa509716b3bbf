#!/usr/bin/env python3
"""Streamed throughput of the Poseidon-transcript verifier against the number of accumulation lanes and batches in flight.
Run on the GPU box: MI_SNARK_VS_LANES=14 python tools/poseidon_stream_probe.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
from snarkos_amd import algorithms as A, lib as L  # noqa: E402

L.init(0)
d = np.load(os.path.join(ROOT, "tests", "golden", "vs_batch_1024_poseidon.npz"))
v = A.VarunaShaped(d["vk"], d["circuits"], transcript="poseidon")
pp, pr = torch.from_numpy(d["proofs"]).cuda(), torch.from_numpy(d["randomizers"]).cuda()
assert v.verify_batch(pp, pr)[0]
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for depth in (4, 8, 12, 14):
    list(v.verify_stream(((pp, pr) for _ in range(16)), acc_depth=depth))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    oks = list(v.verify_stream(((pp, pr) for _ in range(nb)), acc_depth=depth))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert all(oks)
    print("lanes %2d: %.3f ms per batch, %.0f proofs/s" % (depth, 1e3 * dt / nb, 1024 * nb / dt), flush=True)
v.close()
