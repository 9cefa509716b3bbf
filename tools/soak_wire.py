"""Soak of round 3's new kernels under the threaded pipeline (run on the GPU box, bounded wall time): two threads stream
WIRE batches of random sizes with randomly damaged commitments through one verifier context and check every verdict
(vector), a third verifies wire batches synchronously, a fourth runs certificate signatures (the pyref golden records,
replicated) and mi_g1_decompress.  Any mismatch or status ends the run with a non-zero exit code.
    python tools/soak_wire.py [seconds]"""
import json, os, sys, time, threading, random
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snarkos_amd import algorithms as A, lib as L
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
L.init(0)
d = np.load(os.path.join(ROOT, "tests", "golden", "vs_batch_1024.npz"))
v = A.VarunaShaped(d["vk"], d["circuits"], wire=True)
wire_all = v.wire_encode(d["proofs"])
rnd = d["randomizers"]
WB = v.proof_bytes
stop = time.time() + SECONDS
errors, counts = [], {"stream": 0, "sync": 0, "sig": 0, "codec": 0}


def make_batch(rng):
    n = rng.choice([1, 3, 64, 65, 300, 1024, 1024, 1024])
    w = wire_all[: n * WB].copy()
    bad = -1
    if rng.random() < 0.25:
        bad = rng.randrange(n)
        w[bad * WB + 16 + 512 + 48 * rng.randrange(9) + rng.randrange(40)] ^= 1 << rng.randrange(8)
    return torch.from_numpy(w).cuda(), torch.from_numpy(rnd[: n * 32].copy()).cuda(), n, bad


def check(ok, vec, n, bad, where):
    if bad < 0:
        good = ok and vec is None or (ok and vec is not None and vec.all())
    else:
        good = (not ok) and vec is not None and len(vec) == n and vec[bad] == 0 and int(vec.sum()) == n - 1
    if not good:
        errors.append((where, ok, None if vec is None else int(vec.sum()), n, bad))
    return good


def streamer(seed):
    rng = random.Random(seed)
    try:
        while time.time() < stop and not errors:
            batches = [make_batch(rng) for _ in range(rng.randrange(1, 30))]
            res = list(v.verify_stream(((p, r) for p, r, _, _ in batches), verdicts=True, fold_threads=rng.choice([2, 3])))
            for (ok, vec), (_, _, n, bad) in zip(res, batches):
                if not check(ok, vec, n, bad, "stream"):
                    return
            counts["stream"] += len(batches)
    except BaseException as e:
        errors.append(("stream exc", repr(e)))


def syncer(seed):
    rng = random.Random(seed)
    try:
        while time.time() < stop and not errors:
            p, r, n, bad = make_batch(rng)
            ok, vec = v.verify_batch(p, r)
            if not check(ok, vec, n, bad, "sync"):
                return
            counts["sync"] += 1
    except BaseException as e:
        errors.append(("sync exc", repr(e)))


def sig_and_codec(seed):
    R = 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "schnorr.json")))
    mont = lambda h: np.frombuffer((int(h, 16) * (1 << 256) % R).to_bytes(32, "little"), dtype=np.uint8)
    raw = lambda h: np.frombuffer(int(h, 16).to_bytes(32, "little"), dtype=np.uint8)
    recs = np.concatenate([np.concatenate([mont(s["address"][0]), mont(s["address"][1]), mont(s["message"]), raw(s["challenge"]), raw(s["response"]),
                                           mont(s["pk_sig"][0]), mont(s["pk_sig"][1]), mont(s["pr_sig"][0]), mont(s["pr_sig"][1])]) for s in g["signatures"]])
    want = np.array([int(s["valid"]) for s in g["signatures"]], dtype=np.uint8)
    reps = 400
    d_recs = torch.from_numpy(np.tile(recs, reps)).cuda()
    comp = wire_all.reshape(-1, WB)[:, 16 + 512:16 + 512 + 432].reshape(-1, 48)[:4096].copy()
    d_comp = torch.from_numpy(comp).cuda()
    first = None
    try:
        while time.time() < stop and not errors:
            out = A.SchnorrBatch.verify(d_recs).cpu().numpy()
            if not (out.reshape(reps, -1) == want).all():
                errors.append(("schnorr", int(out.sum())))
                return
            counts["sig"] += len(out)
            pts, ok = A.G1Codec.decompress(d_comp, True)
            if not bool(ok.all().item()):
                errors.append(("codec ok",))
                return
            h = pts.cpu().numpy().tobytes()
            if first is None:
                first = h
            elif h != first:
                errors.append(("codec differs",))
                return
            counts["codec"] += len(comp)
    except BaseException as e:
        errors.append(("sig/codec exc", repr(e)))


ths = [threading.Thread(target=streamer, args=(1,)), threading.Thread(target=streamer, args=(2,)), threading.Thread(target=syncer, args=(3,)),
       threading.Thread(target=sig_and_codec, args=(4,))]
t0 = time.time()
for t in ths:
    t.start()
while any(t.is_alive() for t in ths):
    time.sleep(10)
    print("[soak] %.0f s" % (time.time() - t0), counts, flush=True)
for t in ths:
    t.join()
v.close()
print("soak:", counts, "errors:", errors[:3])
sys.exit(1 if errors else 0)
