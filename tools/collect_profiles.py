#!/usr/bin/env python3
"""Copy the judged summaries of a GPU profiling run (tools/profile_round.sh) from gpurun_out/<tag>/ into
profiles/ and derive profiles/pmc_traffic.json (HBM bytes per launch per kernel, read by bench.py).

PMC handling follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are collected in separate passes,
both are in KiB, and on gfx950 FETCH_SIZE counts half the bytes of 16-B-per-lane reads, so it is doubled.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

SHORT = {"pairing_product_kernel": "pairing", "msm_accumulate_kernel": "msm_accumulate", "vs_derive_kernel": "vs_derive",
         "ntt_pass_kernel": "ntt_pass", "msm_bit_sums_kernel": "msm_bit_sums"}


def short(name):
    for k, v in SHORT.items():
        if k in name:
            return v
    return name.split("(")[0].replace("mi::", "").replace("void ", "")


for pat, out in (("stats/*/*_kernel_stats.csv", "%s_rocprofv3_kernel_stats.csv" % tag), ("bench.json", "%s_bench.json" % tag),
                 ("stats_bench.json", "%s_bench_under_rocprofv3.json" % tag)):
    files = sorted(glob.glob(os.path.join(src, pat)), key=os.path.getmtime)
    if files:
        shutil.copy(files[-1], os.path.join(dst, out))  # newest run only

per = {}
for sub, cn in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    agg = collections.defaultdict(list)
    files = sorted(glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    for f in files[-1:]:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == cn:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    per[cn] = {k: sum(v) / len(v) for k, v in agg.items()}
rows = []
traffic = {}
for k in sorted(set(per["FETCH_SIZE"]) | set(per["WRITE_SIZE"])):
    fe, wr = per["FETCH_SIZE"].get(k, 0.0), per["WRITE_SIZE"].get(k, 0.0)
    traffic[k] = (2.0 * fe + wr) * 1024.0
    rows.append((k, fe, wr, traffic[k]))
with open(os.path.join(dst, "%s_pmc_summary.csv" % tag), "w") as f:
    f.write("kernel,FETCH_SIZE_KiB_per_launch(raw),WRITE_SIZE_KiB_per_launch,hbm_bytes_per_launch(2*FETCH+WRITE)\n")
    for r in sorted(rows, key=lambda r: -r[3]):
        f.write("%s,%.3f,%.3f,%.0f\n" % r)
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
print("profiles/ updated from", src)
