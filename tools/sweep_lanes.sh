# streamed throughput (400 and 20 batches) against lanes / product slots (a HIP stream each; 20 in all), product stream priority
cd $GRAFT_REPO_ROOT
for cfg in "10 10 0" "10 10 1" "8 12 0" "8 12 1" "9 11 0" "9 11 1" "11 9 0" "7 13 1"; do
  set -- $cfg
  for k in 400 400 20 20; do
  v=$(MI_SNARK_VS_PRIO=$3 MI_SNARK_VS_LANES=$1 MI_SNARK_VS_SLOTS=$2 timeout -k 10 120 python bench.py --steps $k --warmup 5 --no-extra --no-cpu-baseline --acc-depth $1 --products $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f step %.4f steady %.4f latency %.3f' % (d['value'], d['ms_per_step'], d.get('steady_state_ms_per_step', 0), d['latency_ms_one_batch_unpipelined']))")
  echo "lanes=$1 slots=$2 prio=$3 K=$k -> $v"
  done
done
