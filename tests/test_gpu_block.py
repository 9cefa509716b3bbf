"""-m gpu: one process driving the device list -- mi_init(device_ids, n), mi_vs_verify_block (in-library sharding, RCCL
exchange called from C++, ONE pairing product), the native communicator.  Verdict for verdict against the CPU oracle.
The GPU box has one MI355X: the list [0] and several shards on that one device run everywhere; [0, 1] where a second
GPU is visible."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fixture():
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vs_batch_1024.npz"))
    return d["vk"], d["circuits"], d["proofs"], d["randomizers"]


def _block_8192():
    """BASELINE configs[4]: the committed 1024-proof fixture eight times under eight randomizer sets"""
    vk, circuits, proofs, rnd = _fixture()
    ps, rs = [], []
    for s in range(8):
        r = rnd.copy()
        r[::16] ^= np.uint8(s + 1)
        ps.append(proofs)
        rs.append(r)
    return vk, circuits, np.concatenate(ps), np.concatenate(rs)


def test_device_list_and_abi(mi):
    from snarkos_amd import lib as L
    import torch
    L.init([0])
    assert L.device_count() >= 1 and L.lib().mi_device_id(0) >= 0
    assert L.lib().mi_device_id(99) < 0
    ids = (C.c_int32 * 2)(0, 0)
    assert L.lib().mi_init(ids, 2) == -1          # duplicates
    ids = (C.c_int32 * 1)(torch.cuda.device_count())
    assert L.lib().mi_init(ids, 1) == -2          # no such device
    L.init([0])                                    # idempotent
    n0 = L.device_count()
    L.init(None)
    assert L.device_count() == n0


def test_verify_block_8192_on_the_device_list(orc, mi):
    """the whole 8192-proof block through mi_vs_verify_block (host arrays in, sharded inside the library), all valid and
    with three corrupted proofs; verdict vector against the oracle's per-proof verifier"""
    vk, circuits, proofs, rnd = _block_8192()
    v = mi.VarunaShaped(vk, circuits)
    try:
        ok, verdicts = v.verify_block(proofs, rnd)
        assert ok and verdicts.all() and len(verdicts) == 8192
        bad = proofs.copy()
        idx = [17, 4097, 8191]
        for i in idx:
            bad[i * 2256 + 16 + 512 + 864 + 32 * 5] ^= 1
        ok, verdicts = v.verify_block(bad, rnd)
        want = np.ones(8192, dtype=np.uint8)
        want[idx] = 0
        assert not ok and (verdicts == want).all()
        # the oracle on the shards that hold a corrupted proof (1024 proofs each: seconds)
        for s in (0, 4, 7):
            sl = slice(s * 1024 * 2256, (s + 1) * 1024 * 2256)
            cpu = orc.VSArrays(vk, circuits, bad[sl], rnd[s * 1024 * 32:(s + 1) * 1024 * 32]).verify_cpu(16)
            assert (cpu == want[s * 1024:(s + 1) * 1024]).all()
    finally:
        v.close()


def test_verify_block_device_eight_shards(orc, mi):
    """eight resident shards (all on this box's one GPU: eight host threads, eight lanes, one exchange, one pairing
    product); the partials the block entry point folds are the ones accumulate() returns, which are bit-exact with
    orc.vs_partial_from_oracle; ragged shard sizes and an empty shard"""
    import torch
    vk, circuits, proofs, rnd = _fixture()
    v = mi.VarunaShaped(vk, circuits)
    try:
        cuts = [0, 100, 101, 400, 400, 777, 900, 1000, 1024]  # shard 3 is empty
        shards = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            p = torch.from_numpy(proofs[a * 2256:b * 2256].copy()).cuda() if b > a else torch.zeros(16, dtype=torch.uint8).cuda()
            r = torch.from_numpy(rnd[a * 32:b * 32].copy()).cuda() if b > a else torch.zeros(16, dtype=torch.uint8).cuda()
            shards.append((p, r, b - a))
        live = [(p, r) for p, r, n in shards if n]
        ok, verdicts = v.verify_block_device(live)
        assert ok and verdicts.all() and len(verdicts) == 1024
        # the same partials, one by one, against the oracle
        as_bytes = lambda a: np.frombuffer(np.ascontiguousarray(a).tobytes(), dtype=np.uint8)
        for (a, b), (p, r) in zip([(x, y) for x, y in zip(cuts[:-1], cuts[1:]) if y > x][:3], live[:3]):
            want = as_bytes(orc.vs_partial_from_oracle(orc.VSArrays(vk, circuits, proofs[a * 2256:b * 2256], rnd[a * 32:b * 32]), threads=8))
            assert (v.accumulate(p, r) == want).all()
        bad = proofs.copy()
        bad[500 * 2256 + 16 + 512 + 864 + 32 * 3] ^= 2
        bad[1023 * 2256 + 16 + 512 + 864 + 32 * 20] ^= 1
        live_bad = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            if b > a:
                live_bad.append((torch.from_numpy(bad[a * 2256:b * 2256].copy()).cuda(), torch.from_numpy(rnd[a * 32:b * 32].copy()).cuda()))
        ok, verdicts = v.verify_block_device(live_bad)
        want = orc.VSArrays(vk, circuits, bad, rnd).verify_cpu(16)
        assert not ok and (verdicts == want).all() and want[500] == 0 and want[1023] == 0 and want.sum() == 1022
    finally:
        v.close()


def test_verify_block_two_devices(orc, mi):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    from snarkos_amd import lib as L
    L.init([0, 1])
    assert L.device_count() >= 2
    vk, circuits, proofs, rnd = _block_8192()
    v = mi.VarunaShaped(vk, circuits)
    try:
        ok, verdicts = v.verify_block(proofs, rnd)
        assert ok and verdicts.all()
        bad = proofs.copy()
        bad[6000 * 2256 + 16 + 512 + 864 + 32 * 5] ^= 1
        ok, verdicts = v.verify_block(bad, rnd)
        assert not ok and verdicts.sum() == 8191 and verdicts[6000] == 0
    finally:
        v.close()


def test_native_communicator_world_of_one(orc, mi):
    """RCCL called from the library: unique id, ncclCommInitRank with one rank, all-gather of host records, the
    library's own exchange callback inside the streaming driver, and the in-process communicator over the device list"""
    import torch
    from snarkos_amd import lib as L
    try:
        uid = L.comm_unique_id()
    except L.MiError as e:
        assert e.status == L.MI_ERR_NO_DEVICE
        pytest.skip("no usable RCCL on this host")
    comm = L.comm_init_rank(uid, 0, 1)
    try:
        assert L.lib().mi_comm_world(C.c_uint64(comm)) == 1
        mine = np.arange(608, dtype=np.uint8)
        out = np.zeros(608, dtype=np.uint8)
        L.check(L.lib().mi_comm_all_gather(C.c_uint64(comm), L.ptr(mine), C.c_size_t(608), L.ptr(out)), "all_gather")
        assert (out == mine).all()
        b = orc.VSBatch(0xC011, 40, 3, 0)
        v = mi.VarunaShaped(b.vk, b.circuits)
        try:
            gp, gr = torch.from_numpy(b.proofs).cuda(), torch.from_numpy(b.randomizers).cuda()
            bad = b.proofs.copy()
            bad[9 * 2256 + 16 + 512 + 864 + 32 * 6] ^= 1
            bp = torch.from_numpy(bad).cuda()
            seq = [(gp, gr), (bp, gr), (gp, gr), (gp, gr), (gp, gr), (bp, gr), (gp, gr)]
            want = [True, False, True, True, True, False, True]
            assert list(v.verify_stream(iter(seq), comm=comm)) == want
            assert list(v.verify_stream(iter(seq), comm=comm, exchange_group=3)) == want
        finally:
            v.close()
    finally:
        L.comm_destroy(comm)
    h = C.c_uint64(0)
    assert L.lib().mi_comm_init_local(C.byref(h)) == 0 and L.lib().mi_comm_world(h) == L.device_count()
