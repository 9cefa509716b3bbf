# streamed throughput against the chunk length of pair passes
cd $GRAFT_REPO_ROOT
for t4 in 16 24 32 20 16 24; do
  v=$(MI_SNARK_MSM_SMALL_T4=$t4 timeout -k 10 120 python bench.py --steps 400 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f step %.4f steady %.4f latency %.3f' % (d['value'], d['ms_per_step'], d.get('steady_state_ms_per_step', 0), d['latency_ms_one_batch_unpipelined']))")
  echo "T4=$t4 -> $v"
done
