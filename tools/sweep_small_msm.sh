# streamed throughput against the small-MSM window width bias (interleaved)
cd $GRAFT_REPO_ROOT
for b in -1 0 -1 0 -1 0; do
  v=$(MI_SNARK_MSM_SMALL_C_BIAS=$b timeout -k 10 120 python bench.py --steps 400 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f step %.4f steady %.4f latency %.3f' % (d['value'], d['ms_per_step'], d.get('steady_state_ms_per_step', 0), d['latency_ms_one_batch_unpipelined']))")
  echo "c_bias=$b -> $v"
done
