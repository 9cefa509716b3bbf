"""Worker for tests/test_distributed_cpu.py: one rank of a gloo process group running the sharded
verification protocol (snarkos_amd/sharding.py) with the oracle standing in for the HIP entry points."""
import os
import sys

import numpy as np
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), HERE]

import orc_py as O  # noqa: E402
from snarkos_amd import sharding  # noqa: E402


def msm_mode(rank, world):
    """One MSM with its points split across the ranks: all-gather of 144-byte partial sums, then a local add."""
    import helpers as H

    n = 301  # uneven shards
    bases, scalars = H.rng_points(0x5AD, n), H.rng_scalars(0x5AE, n)
    import pyref as P

    one = O.int_to_limbs(P.FQ_MONT_R, 6)

    def msm(b, s):
        aff = O.msm_pippenger(b, s, 1)
        out = np.zeros(18, dtype=np.uint64)
        if aff.any():
            out[:12], out[12:] = aff, one
        return out

    def sum_partials(parts):
        acc = np.zeros(12, dtype=np.uint64)
        for p in parts:
            if p[12:].any():
                acc = O.g1_add_affine(acc, p[:12]) if acc.any() else p[:12].copy()
        out = np.zeros(18, dtype=np.uint64)
        out[:12], out[12:] = acc, one
        return out

    got = sharding.msm_sharded(bases, scalars, msm, sum_partials, dist)
    want = O.msm_pippenger(bases, scalars, 2)
    assert (got[:12] == want).all() and (got[12:] == one).all()
    print("rank", rank, "msm ok")


def main():
    rank, world, port, corrupt = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if corrupt == -2:
        msm_mode(rank, world)
        dist.destroy_process_group()
        return
    b = O.VSBatch(0xD157, 21, 3, 0)  # 21 proofs: uneven shards
    proofs = b.proofs.copy()
    if corrupt >= 0:
        proofs[corrupt * 2256 + 16 + 512 + 864 + 32 * 5] ^= 1
    seen = []

    def accumulate(p, r):
        seen.append(len(p) // 2256)
        return O.vs_partial_from_oracle(O.VSArrays(b.vk, b.circuits, p, r), threads=2)

    ok = sharding.verify_sharded(proofs, b.randomizers, accumulate, lambda parts: O.vs_finalize_with_oracle(b.vk, parts), dist)
    lo, hi = sharding.shard_bounds(21, world, rank)
    assert seen == [hi - lo]
    # every rank must reach the same verdict, and it must equal the unsharded one
    flags = [None] * world
    dist.all_gather_object(flags, ok)
    assert len(set(flags)) == 1
    whole = O.vs_finalize_with_oracle(b.vk, [O.vs_partial_from_oracle(O.VSArrays(b.vk, b.circuits, proofs, b.randomizers), 2)])
    assert ok == whole == (corrupt < 0), (ok, whole, corrupt)
    dist.destroy_process_group()
    print("rank", rank, "ok", ok)


if __name__ == "__main__":
    main()
