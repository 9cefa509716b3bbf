"""Print the kernel timeline (per HW queue) of a rocprofv3 --kernel-trace CSV between two vs_derive launches."""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "vs_derive" in r["Kernel_Name"]]
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 2]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    name = r["Kernel_Name"].split("(")[0].replace("mi::", "").replace("void ", "")[:28]
    if d > 8 or "derive" in name:
        print("%8.1f %7.1f q%s s%s %s" % (s, d, r["Queue_Id"], r.get("Stream_Id", "?"), name))
d = [int(rows[i]["Start_Timestamp"]) for i in idx]
print("derive-to-derive period (us):", [round((y - x) / 1e3) for x, y in zip(d, d[1:])])
