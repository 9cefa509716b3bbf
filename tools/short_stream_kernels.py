"""Run under rocprofv3 --kernel-trace: one short stream (20 batches) after a warm-up, so that the trace shows when the
device is busy during fill and drain.  The analysis (tools/short_stream_kernels.py --analyse <kernel_trace.csv>) prints,
for the LAST stream of the run, the busy intervals of the big kernels relative to its first kernel."""
import os, sys, time
if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
    import csv
    rows = []
    for r in csv.DictReader(open(sys.argv[2])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # the last stream: kernels after the largest gap
    gaps = [(rows[i + 1][0] - max(x[1] for x in rows[max(0, i - 50):i + 1]), i) for i in range(len(rows) - 1)]
    g, idx = max(gaps)
    last = rows[idx + 1:]
    t0 = last[0][0]
    print("kernels in the last stream:", len(last), " span %.3f ms" % ((max(x[1] for x in last) - t0) / 1e6))
    def span(pattern):
        xs = [(s - t0, e - t0) for s, e, n in last if pattern in n]
        if not xs:
            return
        print("%-28s n=%3d first start %.3f  last end %.3f  total %.3f ms" % (pattern, len(xs), xs[0][0] / 1e6, max(e for _, e in xs) / 1e6, sum(e - s for s, e in xs) / 1e6))
    for pat in ("vs_derive", "msm_digits", "msm_accumulate", "msm_bit_sums", "pairing_lines", "pairing_product"):
        span(pat)
    # per product kernel: start, end
    print("product kernels (start-end ms):", " ".join("%.2f-%.2f" % ((s - t0) / 1e6, (e - t0) / 1e6) for s, e, n in last if "pairing_product" in n))
    print("bit sums kernels (end ms):", " ".join("%.2f" % ((e - t0) / 1e6) for s, e, n in last if "msm_bit_sums" in n))
    sys.exit(0)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snarkos_amd import algorithms as A, lib as L
L.init(0)
d = np.load(os.path.join(ROOT, "tests", "golden", "vs_batch_1024.npz"))
v = A.VarunaShaped(d["vk"], d["circuits"])
p, r = torch.from_numpy(d["proofs"]).cuda(), torch.from_numpy(d["randomizers"]).cuda()
list(v.verify_stream(((p, r) for _ in range(40))))
torch.cuda.synchronize()
time.sleep(0.05)
t0 = time.perf_counter()
ok = list(v.verify_stream(((p, r) for _ in range(20))))
print("20 batches: %.3f ms" % ((time.perf_counter() - t0) * 1e3))
v.close()
