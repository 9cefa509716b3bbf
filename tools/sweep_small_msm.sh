# streamed throughput and one-batch latency against the small-MSM shape (window width bias, chunk length)
cd $GRAFT_REPO_ROOT
for cfg in "-1 16" "0 16" "1 16" "-1 24" "-1 32" "0 24" "0 32" "-2 16"; do
  set -- $cfg
  v=$(MI_SNARK_MSM_SMALL_C_BIAS=$1 MI_SNARK_MSM_SMALL_T=$2 timeout -k 10 120 python bench.py --steps 400 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f step %.4f steady %.4f latency %.3f' % (d['value'], d['ms_per_step'], d.get('steady_state_ms_per_step', 0), d['latency_ms_one_batch_unpipelined']))")
  echo "c_bias=$1 T=$2 -> $v"
done
