#!/bin/bash
# Run on the GPU box (through gpurun): bench line + rocprofv3 kernel stats / kernel trace + PMC traffic passes for the
# three measured workloads (streamed verifier, MSM 2^20, NTT 2^22).  Outputs go to gpurun_out/<tag>/; the summaries are
# copied into profiles/ afterwards (tools/collect_profiles.py <tag>).  PMC passes are separate runs with --kernel-trace
# only (MI355X_MICROARCH.md "HBM"; gpurun refuses --pmc combined with the other trace domains).
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
B="--steps 20 --warmup 5 --no-extra --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $B > $OUT/stats_bench.json 2> $OUT/stats.err || exit 2
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err || exit 3
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err || exit 4
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/batch_stats -- python3 $R/tools/batch_one.py > $OUT/batch.log 2>&1 || exit 8
# the wire form of the same batch (MI_VS_WIRE: decode kernels inside the lane pass) and the certificate signatures
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/wire_stats -- python3 $R/tools/wire_one.py 120 > $OUT/wire.json 2> $OUT/wire.err || exit 9
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/schnorr_stats -- python3 $R/tools/schnorr_one.py > $OUT/schnorr.log 2>&1 || exit 10
for W in "msm20 $R/tools/msm_one.py 20" "ntt22 $R/tools/ntt_one.py 22"; do
  set -- $W
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$1_stats -- python3 $2 $3 > $OUT/$1.log 2>&1 || exit 5
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$1_pmc_fetch -- python3 $2 $3 > /dev/null 2> $OUT/$1_pmc_fetch.err || exit 6
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/$1_pmc_write -- python3 $2 $3 > /dev/null 2> $OUT/$1_pmc_write.err || exit 7
done
find $OUT -name "*.csv" | head -40
