"""Worker for tests/test_distributed_cpu.py::test_host_transport_all_gather: one rank of a communicator made by
mi_comm_init_rank (the unique id arrives through a file); without a GPU the library's transport is the shared-memory
all-gather between the node's processes.  argv: rank world id_file rounds"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snarkos_amd import lib as L  # noqa: E402


def main():
    rank, world, id_file, rounds = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    if rank == 0:
        uid = L.comm_unique_id()
        with open(id_file + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(id_file + ".tmp", id_file)
    else:
        t0 = time.time()
        while not os.path.exists(id_file):
            assert time.time() - t0 < 60
            time.sleep(0.01)
        uid = open(id_file, "rb").read()
    comm = L.comm_init_rank(uid, rank, world)
    assert L.comm_transport(comm) == "host"
    assert L.lib().mi_comm_world(C.c_uint64(comm)) == world
    for r in range(rounds):
        nbytes = 592 * (1 + r % 8)  # groups of 1..8 partials
        mine = np.full(nbytes, (rank * 37 + r) & 0xFF, dtype=np.uint8)
        mine[:8] = np.frombuffer(np.uint64(rank * 1000003 + r).tobytes(), dtype=np.uint8)
        out = np.zeros(world * nbytes, dtype=np.uint8)
        L.check(L.lib().mi_comm_all_gather(C.c_uint64(comm), L.ptr(mine), C.c_size_t(nbytes), L.ptr(out)), "mi_comm_all_gather")
        got = out.reshape(world, nbytes)
        for k in range(world):
            assert int(np.frombuffer(got[k, :8].tobytes(), dtype=np.uint64)[0]) == k * 1000003 + r, (rank, r, k)
            assert (got[k, 8:] == ((k * 37 + r) & 0xFF)).all()
        if rank == r % world:
            time.sleep(0.002)  # a straggler: the others must wait for it, and must not overwrite what it has not read
    s = L.stats_read()
    assert s["exchange_calls"] == rounds and s["exchange_nanos"] > 0
    L.comm_destroy(comm)
    print("rank", rank, "ok")


if __name__ == "__main__":
    main()
