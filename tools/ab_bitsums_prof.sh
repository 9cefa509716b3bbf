# kernel durations of one batch alone and of the streamed bench, for both bit-sum reductions
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/absum
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lpg in 0 1; do
  export MI_SNARK_BITSUM_LPG=$lpg
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/one$lpg -- python3 $R/tools/batch_one.py > $OUT/one$lpg.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/str$lpg -- python3 $R/bench.py --steps 200 --warmup 5 --no-extra --no-cpu-baseline > $OUT/str$lpg.json 2> $OUT/str$lpg.err || exit 2
done
for d in one0 one1 str0 str1; do echo "== $d"; f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); cut -d, -f1-4,6,7 $f | sed 's/(.*)"/"/' | head -9; done
