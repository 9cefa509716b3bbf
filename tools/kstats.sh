#!/bin/bash
# Run on the GPU box: rocprofv3 kernel stats of one tool script, the per-kernel averages printed.  tools/kstats.sh <tag> <script> [args]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/r03/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/$@ > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        print("%-46s calls %5s  avg %9.1f us  min %9.1f us" % (r['Name'][:46], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
