# streamed throughput against the number of accumulation lanes and product slots (each has its own HIP stream)
cd $GRAFT_REPO_ROOT
for cfg in "12 8" "16 8" "20 8" "24 8" "16 6" "20 6" "16 4" "20 4"; do
  set -- $cfg
  v=$(MI_SNARK_VS_LANES=$1 MI_SNARK_VS_SLOTS=$2 timeout -k 10 120 python bench.py --steps 400 --warmup 5 --no-extra --no-cpu-baseline --acc-depth $1 --products $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f step %.4f steady %.4f latency %.3f' % (d['value'], d['ms_per_step'], d.get('steady_state_ms_per_step', 0), d['latency_ms_one_batch_unpipelined']))")
  echo "lanes=$1 slots=$2 -> $v"
done
