"""Where a short stream's time goes: from a rocprofv3 --kernel-trace CSV of `bench.py --steps 20`, the LAST group of 20
vs_derive-led batches' kernels: per kernel name the first start, last end and summed duration relative to the first
derive of the timed region, plus how many kernels ran side by side over time (10 buckets)."""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
S = lambda r: int(r["Start_Timestamp"]); E = lambda r: int(r["End_Timestamp"])
# regions: gaps > 2 ms between kernels split the run into streams (warm-up, timed, steady, bracketed ...)
regions, cur = [], [rows[0]]
for r in rows[1:]:
    if S(r) - max(E(x) for x in cur[-50:]) > 2_000_000:
        regions.append(cur); cur = []
    cur.append(r)
regions.append(cur)
which = int(sys.argv[2]) if len(sys.argv) > 2 else None
for i, reg in enumerate(regions):
    nd = sum(1 for r in reg if "vs_derive" in r["Kernel_Name"])
    print("region %d: %d kernels, %d derive launches, %.2f ms" % (i, len(reg), nd, (max(E(r) for r in reg) - S(reg[0])) / 1e6))
if which is None:
    sys.exit(0)
reg = regions[which]
t0 = S(reg[0])
agg = collections.OrderedDict()
for r in reg:
    n = r["Kernel_Name"].split("(")[0].replace("mi::", "").replace("void ", "")[:34]
    a = agg.setdefault(n, [1e18, 0, 0, 0])
    a[0] = min(a[0], S(r) - t0); a[1] = max(a[1], E(r) - t0); a[2] += E(r) - S(r); a[3] += 1
for n, a in agg.items():
    print("%-34s first %7.1f us  last end %7.1f us  sum %8.1f us  x%d  avg %6.1f" % (n, a[0] / 1e3, a[1] / 1e3, a[2] / 1e3, a[3], a[2] / 1e3 / a[3]))
end = max(E(r) for r in reg) - t0
for b in range(20):
    lo, hi = end * b / 20, end * (b + 1) / 20
    busy = sum(max(0, min(E(r) - t0, hi) - max(S(r) - t0, lo)) for r in reg) / (hi - lo)
    names = collections.Counter(r["Kernel_Name"].split("(")[0].replace("mi::", "").replace("void ", "")[:20] for r in reg if S(r) - t0 < hi and E(r) - t0 > lo)
    print("%5.2f-%5.2f ms: %.1f kernels in flight  %s" % (lo / 1e6, hi / 1e6, busy, dict(names.most_common(4))))
