#!/bin/bash
# Run on the GPU box (through gpurun): bench line + rocprofv3 kernel stats + PMC traffic passes.
# Outputs go to gpurun_out/; the summaries are copied into profiles/ afterwards (tools/collect_profiles.py).
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-extra --no-cpu-baseline > $OUT/stats_bench.json 2> $OUT/stats.err || exit 2
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err || exit 3
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err || exit 4
find $OUT -name "*.csv" | head -20
ls -la $OUT/stats/*/ 2>/dev/null | head
