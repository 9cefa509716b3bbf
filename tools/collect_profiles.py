#!/usr/bin/env python3
"""Copy the judged summaries of a GPU profiling run (tools/profile_round.sh <tag>) from gpurun_out/<tag>/ into
profiles/ and derive
  profiles/<tag>_pmc_summary.csv   HBM bytes per launch per kernel, all three workloads
  profiles/pmc_traffic.json        the same as a table bench.py reads for roofline.traffic
  profiles/<tag>_overlap.json      how many pairing products run side by side in the streamed region, from the kernel
                                   trace (reconciles "a ~1 ms kernel inside a ~0.2 ms step")

PMC handling follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are collected in separate passes, both are
in KiB, and on gfx950 FETCH_SIZE counts half the bytes of 16-B-per-lane reads, so it is doubled.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

SHORT = {"pairing_product_multi_kernel": "pairing_multi", "pairing_product_kernel": "pairing", "msm_accumulate_kernel": "msm_accumulate", "vs_derive_kernel": "vs_derive",
         "ntt_pass_kernel": "ntt_pass", "ntt_lds_pass_kernel": "ntt_lds_pass", "msm_bit_sums_queue_kernel": "msm_bit_sums_queue"}


def short(name):
    for k, v in SHORT.items():
        if k in name:
            return v
    return name.split("(")[0].split("<")[0].replace("mi::", "").replace("void ", "").strip()


def newest(pattern):
    files = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)
    return files[-1] if files else None


for pat, out in (("stats/*/*_kernel_stats.csv", "%s_rocprofv3_kernel_stats.csv" % tag), ("bench.json", "%s_bench.json" % tag),
                 ("stats_bench.json", "%s_bench_under_rocprofv3.json" % tag),
                 ("msm20_stats/*/*_kernel_stats.csv", "%s_msm20_kernel_stats.csv" % tag),
                 ("ntt22_stats/*/*_kernel_stats.csv", "%s_ntt22_kernel_stats.csv" % tag),
                 ("batch_stats/*/*_kernel_stats.csv", "%s_one_batch_kernel_stats.csv" % tag),
                 ("wire_stats/*/*_kernel_stats.csv", "%s_wire_batch_kernel_stats.csv" % tag), ("wire.json", "%s_wire_batch.json" % tag),
                 ("schnorr_stats/*/*_kernel_stats.csv", "%s_schnorr_kernel_stats.csv" % tag)):
    f = newest(pat)
    if f:
        shutil.copy(f, os.path.join(dst, out))

# ---- PMC traffic per launch
rows, traffic = [], {}
for work, pf, pw in (("verifier", "pmc_fetch", "pmc_write"), ("msm20", "msm20_pmc_fetch", "msm20_pmc_write"),
                     ("ntt22", "ntt22_pmc_fetch", "ntt22_pmc_write")):
    per = {}
    for sub, cn in ((pf, "FETCH_SIZE"), (pw, "WRITE_SIZE")):
        agg = collections.defaultdict(list)
        f = newest(os.path.join(sub, "*", "*_counter_collection.csv"))
        if f:
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == cn:
                    agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        per[cn] = {k: sum(v) / len(v) for k, v in agg.items()}
    for k in sorted(set(per["FETCH_SIZE"]) | set(per["WRITE_SIZE"])):
        fe, wr = per["FETCH_SIZE"].get(k, 0.0), per["WRITE_SIZE"].get(k, 0.0)
        t = (2.0 * fe + wr) * 1024.0
        rows.append((work, k, fe, wr, t))
        traffic["%s:%s" % (work, k)] = t
        if work == "verifier":
            traffic[k] = t  # what bench.py's roofline.traffic looks up
with open(os.path.join(dst, "%s_pmc_summary.csv" % tag), "w") as f:
    f.write("workload,kernel,FETCH_SIZE_KiB_per_launch(raw),WRITE_SIZE_KiB_per_launch,hbm_bytes_per_launch(2*FETCH+WRITE)\n")
    for r in sorted(rows, key=lambda r: (r[0], -r[4])):
        f.write("%s,%s,%.3f,%.3f,%.0f\n" % r)
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)

# ---- overlap of pairing products in the streamed region
f = newest("stats/*/*_kernel_trace.csv")
if f:
    ev, durs, first, last, n_products = [], [], None, None, 0
    for r in csv.DictReader(open(f)):
        if "pairing_product_kernel" not in r["Kernel_Name"] and "pairing_product_multi_kernel" not in r["Kernel_Name"]:
            continue
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        products = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])  # a workgroup per product (one, or up to four under one ticket)
        ev.append((s, products))
        ev.append((e, -products))
        durs.append(e - s)
        n_products += products
        first = s if first is None else min(first, s)
        last = e if last is None else max(last, e)
    ev.sort()
    cur, mx, area, busy, prev = 0, 0, 0, 0, None
    for t, d in ev:
        if prev is not None and cur > 0:
            area += cur * (t - prev)
            busy += t - prev
        cur += d
        mx = max(mx, cur)
        prev = t
    n = len(durs)
    out = {"kernel": "pairing_product_kernel + pairing_product_multi_kernel", "launches": n, "products": n_products,
           "avg_kernel_us": sum(durs) / n / 1e3 if n else None,
           "max_concurrent_products": mx, "mean_concurrent_products_while_any_runs": area / busy if busy else None,
           "wall_us_with_a_product_running": busy / 1e3, "wall_us_per_product": busy / n_products / 1e3 if n_products else None,
           "cu_busy_us_per_product": area / n_products / 1e3 if n_products else None,
           "note": "one workgroup (3 waves, one CU) per product, one to four products per launch: a step's wall time is the "
                   "kernel's duration divided by the number of products in flight; trace of `bench.py --steps 20 --warmup 5` "
                   "(warm-up, timed, steady-state and event-bracketed passes together)"}
    json.dump(out, open(os.path.join(dst, "%s_overlap.json" % tag), "w"), indent=1)
print("profiles/ updated from", src)
