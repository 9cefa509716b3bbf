"""Canonical BLS12-377 scalars uniform in [0, r) as an (n, 4) int64 torch tensor (the bytes of mi_bigint256): rejection
sampling on 253 bits ("scalars uniform in [0, r)").  Used by bench.py and the profiling tools."""
import numpy as np
import torch

R_LIMBS = [0x0A11800000000001, 0x59AA76FED0000001, 0x60B44D1E5C37B001, 0x12AB655E9A2CA556]


def uniform_scalars(n, seed):
    g = np.random.default_rng(seed)
    out = np.zeros((n, 4), dtype=np.uint64)
    todo = np.arange(n)
    while len(todo):
        a = g.integers(0, 1 << 63, size=(len(todo), 4), dtype=np.uint64) * np.uint64(2) + g.integers(0, 2, size=(len(todo), 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 61) - 1)  # 253 bits
        lt = np.zeros(len(todo), dtype=bool)
        eq = np.ones(len(todo), dtype=bool)
        for k in (3, 2, 1, 0):
            lt |= eq & (a[:, k] < np.uint64(R_LIMBS[k]))
            eq &= a[:, k] == np.uint64(R_LIMBS[k])
        out[todo[lt]] = a[lt]
        todo = todo[~lt]
    return torch.from_numpy(out.view(np.int64))
