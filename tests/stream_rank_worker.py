"""One rank of the STREAMED sharded verifier on the HIP path (csrc/stream.hip's grouped exchange), for
tests/test_gpu_two_ranks.py: argv = rank world id_file exchange(native|gloo) group port out_json.
Every rank holds its shard of each of 8 batches (a 1024-proof fixture cut into `world` contiguous shards); batch 3 carries
one corrupted proof in the LAST rank's shard.  Every rank must report all_ok = [1,1,1,0,1,1,1,1]; the last rank's verdict
vector of batch 3 names exactly that proof, the other ranks' vectors are all ones (each rank names the invalid proofs of
its own shard)."""
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
os.environ.setdefault("MI_SNARK_VS_LANES", "4")   # several ranks share one device in this rehearsal: few streams each
os.environ.setdefault("MI_SNARK_VS_SLOTS", "3")
import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snarkos_amd import algorithms as A, lib as L, sharding  # noqa: E402


def main():
    rank, world, id_file, exchange, group, port, out_json = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4],
                                                             int(sys.argv[5]), sys.argv[6], sys.argv[7])
    torch.cuda.set_device(0)
    L.init(0)
    d = np.load(os.path.join(ROOT, "tests", "golden", "vs_batch_1024.npz"))
    n = 256  # proofs per batch over all ranks
    lo, hi = sharding.shard_bounds(n, world, rank)
    PB = L.VS_PROOF_BYTES
    shard = d["proofs"][lo * PB:hi * PB].copy()
    rnd = d["randomizers"][lo * 32:hi * 32].copy()
    bad = shard.copy()
    bad_index = (hi - lo) // 2
    if rank == world - 1:
        bad[bad_index * PB + 16 + 512 + 864 + 32 * 5] ^= 1
    v = A.VarunaShaped(d["vk"], d["circuits"])
    dp, db, dr = torch.from_numpy(shard).cuda(), torch.from_numpy(bad).cuda(), torch.from_numpy(rnd).cuda()
    comm, gather = None, None
    if exchange == "native":
        if rank == 0:
            uid = L.comm_unique_id()
            with open(id_file + ".tmp", "wb") as f:
                f.write(uid)
            os.rename(id_file + ".tmp", id_file)
        else:
            t0 = time.time()
            while not os.path.exists(id_file):
                assert time.time() - t0 < 120
                time.sleep(0.01)
            uid = open(id_file, "rb").read()
        comm = L.comm_init_rank(uid, rank, world)
        transport = L.comm_transport(comm)
    else:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
        dist.init_process_group("gloo", rank=rank, world_size=world)
        transport = "gloo callback"

        def gather(part):
            part = np.ascontiguousarray(part, dtype=np.uint8)
            return sharding.gather_bytes(part, part.size, dist, None).reshape(world, -1, L.VS_PARTIAL_BYTES)

    batches = [(db if k == 3 else dp, dr) for k in range(8)]
    s0 = L.stats_read()
    t0 = time.perf_counter()
    res = list(v.verify_stream(iter(batches), gather, depth=3, acc_depth=4, fold_threads=2, exchange_group=group, comm=comm, verdicts=True))
    dt = time.perf_counter() - t0
    s1 = L.stats_read()
    oks = [int(ok) for ok, _ in res]
    vec = res[3][1]
    want = np.ones(hi - lo, dtype=np.uint8)
    if rank == world - 1:
        want[bad_index] = 0
    out = {"rank": rank, "oks": oks, "verdicts_match": bool(vec is not None and (vec == want).all()), "transport": transport,
           "ms_per_batch": 1e3 * dt / 8, "exchange_calls": s1["exchange_calls"] - s0["exchange_calls"],
           "exchange_wait_ms_per_call": 1e-6 * (s1["exchange_nanos"] - s0["exchange_nanos"]) / max(1, s1["exchange_calls"] - s0["exchange_calls"])}
    v.close()
    with open(out_json, "w") as f:
        json.dump(out, f)
    print(json.dumps(out))
    assert oks == [1, 1, 1, 0, 1, 1, 1, 1] and out["verdicts_match"], out


if __name__ == "__main__":
    main()
