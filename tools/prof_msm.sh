#!/bin/bash
# rocprofv3 kernel stats of one large MSM (run on the GPU box): tools/prof_msm.sh LOG_N
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/msm_prof_$1
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/msm_one.py $1 > $OUT/run.log 2>&1 || { echo "profile run failed"; tail -3 $OUT/run.log; exit 2; }
f=$(ls $OUT/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] || { echo "no stats file"; ls -R $OUT | head; exit 3; }
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(r["Name"][:60].ljust(60), r["Calls"].rjust(5), "%10.1f us avg" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
PY
