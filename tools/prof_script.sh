#!/bin/bash
# rocprofv3 kernel stats of a python tool script (run on the GPU box): tools/prof_script.sh tools/x.py [args]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_script
rm -rf $OUT; mkdir -p $OUT
S=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $S "$@" > $OUT/run.log 2>&1 || { echo "profile run failed"; tail -3 $OUT/run.log; exit 2; }
f=$(ls $OUT/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] || { echo "no stats file"; exit 3; }
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(r["Name"][:60].ljust(60), r["Calls"].rjust(5), "%10.1f us avg" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
PY
