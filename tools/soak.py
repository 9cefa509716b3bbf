"""Soak test for the threaded pipeline (ad-hoc; run on the GPU box): several host threads hammer the library at once --
streamed verification with random batch sizes and randomly corrupted batches, synchronous verify_batch, MSM and NTT
calls -- for a fixed wall time; every verdict / result is checked (prefixes of the fixture batch are valid, a flipped
evaluation bit makes a batch invalid; MSM / NTT results must repeat bit-exactly)."""
import os, sys, time, threading, random
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snarkos_amd import algorithms as A, lib as L
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
L.init(0)
d = np.load(os.path.join(ROOT, "tests", "golden", "vs_batch_1024.npz"))
proofs, rnd = d["proofs"], d["randomizers"]
v = A.VarunaShaped(d["vk"], d["circuits"])
stop = time.time() + SECONDS
errors, counts = [], {"stream": 0, "sync": 0, "msm": 0, "ntt": 0}

def make_batch(rng):
    n = rng.choice([1, 2, 7, 64, 300, 1024, 1024, 1024])
    p = proofs[: n * 2256].copy()
    bad = rng.random() < 0.2
    if bad:
        i = rng.randrange(n)
        p[i * 2256 + 16 + 512 + 864 + 32 * rng.randrange(21)] ^= 1 << rng.randrange(8)
    return torch.from_numpy(p).cuda(), torch.from_numpy(rnd[: n * 32].copy()).cuda(), (not bad)

def streamer(seed):
    rng = random.Random(seed)
    try:
        while time.time() < stop:
            batches = [make_batch(rng) for _ in range(rng.randrange(1, 40))]
            got = list(v.verify_stream(((p, r) for p, r, _ in batches), fold_threads=rng.choice([1, 2, 3])))
            want = [ok for _, _, ok in batches]
            if got != want:
                errors.append(("stream", got, want)); return
            counts["stream"] += len(batches)
    except BaseException as e:
        errors.append(("stream exc", repr(e)))

def syncer(seed):
    rng = random.Random(seed)
    try:
        while time.time() < stop:
            p, r, ok = make_batch(rng)
            got, verdicts = v.verify_batch(p, r)
            if got != ok or (ok and not verdicts.all()) or (not ok and verdicts.sum() != len(verdicts) - 1):
                errors.append(("sync", got, ok, int(verdicts.sum()))); return
            counts["sync"] += 1
    except BaseException as e:
        errors.append(("sync exc", repr(e)))

def msm_ntt(seed):
    g = np.random.default_rng(seed)
    n = 3000
    import ctypes as C
    pts = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
    L.check(L.lib().mi_dbg_gen_points_device(C.c_uint64(seed), C.c_size_t(n), L.dptr(pts), L.stream_ptr()), "gen")
    torch.cuda.synchronize()
    sc = g.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64((1 << 58) - 1)
    scd = torch.from_numpy(sc.view(np.int64)).cuda()
    ref = A.VariableBase.msm(pts, scd).copy()
    a = sc[:2048].copy(); dom = A.EvaluationDomain(2048); fref = dom.fft(a).copy()
    try:
        while time.time() < stop:
            if not (A.VariableBase.msm(pts, scd) == ref).all():
                errors.append(("msm mismatch",)); return
            counts["msm"] += 1
            f = dom.fft(a)
            if not (f == fref).all() or not (dom.ifft(f) == a).all():
                errors.append(("ntt mismatch",)); return
            counts["ntt"] += 1
    except BaseException as e:
        errors.append(("msm/ntt exc", repr(e)))

# one streamer at a time per context (verify_stream owns the lanes); the others share the library lock
ts = [threading.Thread(target=streamer, args=(1,)), threading.Thread(target=msm_ntt, args=(3,)), threading.Thread(target=msm_ntt, args=(4,))]
for t in ts: t.start()
for t in ts: t.join(timeout=SECONDS + 120)
alive = [t for t in ts if t.is_alive()]
# second phase: synchronous verify_batch from two threads + MSM
stop = time.time() + SECONDS / 3
ts = [threading.Thread(target=syncer, args=(5,)), threading.Thread(target=syncer, args=(6,)), threading.Thread(target=msm_ntt, args=(7,))]
for t in ts: t.start()
for t in ts: t.join(timeout=SECONDS + 120)
alive += [t for t in ts if t.is_alive()]
print("counts", counts, "errors", errors[:3], "hung threads", len(alive))
sys.exit(1 if errors or alive else 0)
