#!/usr/bin/env python3
"""Write tests/golden/vs_batch_1024.npz (SHA-256 transcript) and vs_batch_1024_poseidon.npz (Poseidon transcript): 1024
valid VarunaShaped-v1 proofs over 8 circuits (seed 0xA1E00004), and vs2_batch_1024.npz (VarunaShaped-v2, seed 0xA1E00005), made by the oracle's synthetic prover (toy SRS).  bench.py and the full-size GPU tests load
this file, so the measured path never runs the prover.  Run:  python oracle/gen_vs_fixture.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import orc_py as O  # noqa: E402

for name, kind in (("vs_batch_1024.npz", O.FS_SHA256), ("vs_batch_1024_poseidon.npz", O.FS_POSEIDON)):
    # the same seed under both Fiat-Shamir kinds: same keys, inputs and commitments; evaluations and witnesses differ
    b = O.VSBatch(0xA1E00004, 1024, 8, 0, transcript=kind)
    assert b.expected.all() and b.verify_cpu(8).all() and b.verify_aggregated(8)[0]
    out = os.path.join(os.path.dirname(HERE), "tests", "golden", name)
    np.savez(out, vk=b.vk, circuits=b.circuits, proofs=b.proofs, randomizers=b.randomizers, seed=np.uint64(b.seed))
    print("wrote", out, os.path.getsize(out), "bytes")

# VarunaShaped-v2 (default shape: hiding, two degree-bound classes, the v1 query table): tests/golden/vs2_batch_1024.npz
b = O.VS2Batch(0xA1E00005, 1024, 8, 0)
assert b.expected.all() and b.verify_cpu(8).all() and b.verify_aggregated(8)[0]
out = os.path.join(os.path.dirname(HERE), "tests", "golden", "vs2_batch_1024.npz")
np.savez(out, vk=b.vk, ext=b.ext, shape=b.shape, circuits=b.circuits, proofs=b.proofs, randomizers=b.randomizers, seed=np.uint64(b.seed))
print("wrote", out, os.path.getsize(out), "bytes")
