#!/usr/bin/env python3
"""Where the time of a rejected batch goes: the 1024-proof fixture with 11 corrupted proofs through mi_vs_verify_batch_device,
wall time and the library's own event timers for the fallback kernels.  Run on the GPU box: python tools/reject_profile.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snarkos_amd import algorithms as A, lib as L  # noqa: E402

L.init(0)
d = np.load(os.path.join(ROOT, "tests", "golden", "vs_batch_1024.npz"))
v = A.VarunaShaped(d["vk"], d["circuits"])
bad = d["proofs"].copy()
idx = np.arange(5, 1024, 101)
for i in idx:
    bad[i * L.VS_PROOF_BYTES + 16 + 512 + 864 + 32 * 7] ^= 1
dp, db, dr = (torch.from_numpy(x).cuda() for x in (d["proofs"], bad, d["randomizers"]))
want = np.ones(1024, dtype=np.uint8)
want[idx] = 0
for name, x in (("valid", dp), ("1pct_corrupted", db)):
    v.verify_batch(x, dr)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        ok, verdicts = v.verify_batch(x, dr)
        ts.append(time.perf_counter() - t0)
    print("%-16s wall ms: %s  all_ok=%s" % (name, " ".join("%.2f" % (1e3 * t) for t in ts), ok))
    if name != "valid":
        assert os.environ.get("REJECT_NOCHECK") or (verdicts == want).all()
L.prof_enable(2)
for k in ("vs_terms", "vs_sums", "vs_per_proof", "vs_groups", "vs_derive"):
    L.prof_read(k)
for _ in range(3):
    v.verify_batch(db, dr)
for k in ("vs_derive", "vs_terms", "vs_sums", "vs_groups", "vs_per_proof"):
    ms, launches, _ = L.prof_read(k)
    if launches:
        print("%-14s %.3f ms x %d" % (k, ms / launches, launches))
L.prof_enable(0)
v.close()
