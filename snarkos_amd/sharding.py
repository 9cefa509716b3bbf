"""Multi-GPU sharding of one block-validation batch.

Proofs are independent until the final accumulator, so a batch of N proofs is cut into contiguous shards,
one per rank (one process per GPU).  Each rank folds its shard into a 592-byte `mi_vs_partial`
(two G1 accumulators + counts); the only exchange is an all-gather of those partials (RCCL over xGMI on the
GPU box, gloo in the CPU tests), after which any rank adds the partials and runs the single pairing product.
RCCL has no elliptic-curve reduction operator, hence gather + local add instead of a reduce.

The arithmetic is injected (`accumulate`, `finalize`), so the protocol is the same object on both tiers: the
product passes the HIP library's entry points, the CPU tests pass the oracle's.
"""
import numpy as np

PARTIAL_BYTES = 592
PROOF_BYTES = 2256
G1_JAC_BYTES = 144


def shard_bounds(n_items, world, rank):
    """Contiguous shard [lo, hi) of rank: sizes differ by at most one, earlier ranks take the remainder."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_bytes(blob, nbytes, dist=None, device=None):
    """All-gather one fixed-size byte record per rank -> (world, nbytes) uint8 array on every rank."""
    blob = np.ascontiguousarray(blob).view(np.uint8).reshape(nbytes)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return blob.reshape(1, nbytes)
    import torch

    world = dist.get_world_size()
    mine = torch.from_numpy(blob.copy())
    if device is not None:
        mine = mine.to(device)
    out = torch.empty(world * nbytes, dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(out, mine)
    return out.cpu().numpy().reshape(world, nbytes)


def gather_partials(partial, dist=None, device=None):
    """All-gather one 592-byte partial per rank -> (world, 592) uint8 array on every rank."""
    return gather_bytes(np.ascontiguousarray(partial, dtype=np.uint8), PARTIAL_BYTES, dist, device)


def msm_sharded(bases, scalars, msm, sum_partials, dist=None, device=None, pre_sharded=False):
    """One large MSM with its POINTS split across ranks (windows are not split).

    bases (n, 12) / scalars (n, 4) uint64: the whole problem on every rank (rank r takes shard r), or this rank's
    shard with pre_sharded=True.  msm(bases, scalars) -> (18,) uint64 normalised Jacobian result;
    sum_partials((world, 18)) -> (18,).  The only exchange is an all-gather of one 144-byte point per rank.
    Returns the full result (identical on every rank).
    """
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    if not pre_sharded:
        n = min(len(bases), len(scalars))
        lo, hi = shard_bounds(n, world, rank)
        bases, scalars = bases[lo:hi], scalars[lo:hi]
    part = np.ascontiguousarray(msm(bases, scalars), dtype=np.uint64).reshape(G1_JAC_BYTES // 8)
    parts = gather_bytes(part, G1_JAC_BYTES, dist, device)
    return np.asarray(sum_partials(parts.view(np.uint64).reshape(world, G1_JAC_BYTES // 8)), dtype=np.uint64)


def verify_sharded(proofs, randomizers, accumulate, finalize, dist=None, device=None, pre_sharded=False):
    """Verify one batch across the ranks of `dist`.

    proofs / randomizers: the whole batch (every rank holds it; rank r works on shard r), or with
    pre_sharded=True this rank's shard only.  accumulate(proofs_shard, randomizers_shard) -> 592-byte partial;
    finalize((world, 592) partials) -> bool.  Returns all_ok (identical on every rank).
    """
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    if not pre_sharded:
        n = len(proofs) // PROOF_BYTES
        lo, hi = shard_bounds(n, world, rank)
        proofs = proofs[lo * PROOF_BYTES:hi * PROOF_BYTES]
        randomizers = randomizers[lo * 32:hi * 32]
    part = accumulate(proofs, randomizers)
    return bool(finalize(gather_partials(part, dist, device)))
