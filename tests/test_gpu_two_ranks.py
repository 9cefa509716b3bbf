"""-m gpu: the N > 1 STREAMED path on the HIP library (the reference node's counterpart is one process,
reference cli/src/commands/start.rs:616-642): two ranks as two fresh processes on the one GPU of the box, each
streaming its shards of 8 batches through mi_vs_stream_* with the grouped partial exchange of csrc/stream.hip --
 * natively (mi_comm_init_rank + mi_vs_exchange_rccl: two ranks on one device cannot use RCCL, so the library's vote
   picks its shared-memory transport), exchange groups of 1 and 3 batches;
 * through the torch.distributed (gloo) callback.
One corrupted proof sits in rank 1's shard of batch 3: both ranks must reject exactly that batch, and rank 1's verdict
vector must name the proof.  The ranks are children of tests/spawner.py, which never touched the GPU."""
import json
import os
import socket
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


@pytest.mark.parametrize("exchange,group", [("native", 1), ("native", 3), ("gloo", 2)])
def test_two_ranks_stream_on_one_gpu(spawn, tmp_path, exchange, group):
    world = 2
    idf, port = str(tmp_path / "comm.id"), _port()
    outs = [str(tmp_path / ("rank%d.json" % r)) for r in range(world)]
    cmds = [[sys.executable, os.path.join(HERE, "stream_rank_worker.py"), str(r), str(world), idf, exchange, str(group), port, outs[r]]
            for r in range(world)]
    rc, out = spawn(cmds, timeout=400)
    assert rc == [0] * world, "\n".join(out)
    reps = [json.load(open(o)) for o in outs]
    for r in reps:
        assert r["oks"] == [1, 1, 1, 0, 1, 1, 1, 1] and r["verdicts_match"]
        assert r["transport"] == ("host" if exchange == "native" else "gloo callback")
        if exchange == "native":
            assert r["exchange_calls"] == -(-8 // group)  # one all-gather per group of batches
