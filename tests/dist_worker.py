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


def main():
    rank, world, port, corrupt = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
