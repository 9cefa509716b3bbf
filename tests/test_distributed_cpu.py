"""CPU tier: the N > 1 path (contiguous shards, all-gather of 592-byte partials, one finalize) with
world_size 2 and 3 over gloo.  The oracle stands in for the HIP accumulate/finalize entry points; the
sharding protocol object (snarkos_amd/sharding.py) is the one bench.py uses over RCCL."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


def _run(world, corrupt):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(r), str(world), port, str(corrupt)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_shard_bounds():
    from snarkos_amd.sharding import shard_bounds
    for n in (0, 1, 7, 8, 1024, 8191):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(n, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_valid_batch(world):
    _run(world, -1)


def test_sharded_batch_with_a_bad_proof():
    _run(2, 13)  # the corrupted proof lands in rank 1's shard; both ranks must reject


def test_sharded_msm():
    _run(3, -2)  # points split 101/100/100 across three ranks, 144-byte partial sums all-gathered


@pytest.mark.parametrize("world", [2, 3])
def test_host_transport_all_gather(world, tmp_path):
    """the library's own exchange between rank-per-process communicators when RCCL does not serve them (here: no GPU):
    mi_comm_unique_id / mi_comm_init_rank / mi_comm_all_gather over the shared-memory transport, 60 rounds of 1..8
    partials with stragglers"""
    idf = str(tmp_path / "comm.id")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "comm_worker.py"), str(r), str(world), idf, "60"],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
