"""Single-rank RCCL process group: the sharded stream path (all-gather of partials on the finisher thread) with
world_size 1 -- checks that collectives issued from that thread work and what they cost per batch (ad-hoc probe)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
import numpy as np, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snarkos_amd import algorithms as A, lib as L, sharding
backend = sys.argv[1] if len(sys.argv) > 1 else "nccl"
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
if backend == "nccl":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
else:
    dist.init_process_group(backend, rank=0, world_size=1)
L.init(0)
d = np.load(os.path.join(ROOT, "tests", "golden", "vs_batch_1024.npz"))
v = A.VarunaShaped(d["vk"], d["circuits"])
p = torch.from_numpy(d["proofs"]).cuda(); r = torch.from_numpy(d["randomizers"]).cuda()
gdev = dev if backend == "nccl" else None
def gather(part):  # (count, 592) -> (world = 1, count, 592), through a real collective
    torch.cuda.set_device(0)
    mine = torch.from_numpy(np.ascontiguousarray(part).reshape(-1).copy())
    mine = mine.to(gdev) if gdev is not None else mine
    out = torch.empty_like(mine)
    dist.all_gather_into_tensor(out, mine)
    return out.cpu().numpy().reshape(1, -1, 592)
GROUP = int(os.environ.get("PROBE_GROUP", "1"))
for g in (None, gather):
    list(v.verify_stream(((p, r) for _ in range(10)), g, exchange_group=GROUP))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    oks = list(v.verify_stream(((p, r) for _ in range(200)), g, exchange_group=GROUP))
    torch.cuda.synchronize()
    assert all(oks)
    print(backend, "gather" if g else "no gather", "%.3f ms/batch" % (1e3 * (time.perf_counter() - t0) / 200))
dist.destroy_process_group()
