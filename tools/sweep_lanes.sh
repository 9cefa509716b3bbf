# streamed throughput (400 and 20 batches) against lanes / product slots now that a ticket carries up to four products
cd $GRAFT_REPO_ROOT
for cfg in "10 10" "12 8" "13 7" "14 6" "12 6" "14 4"; do
  set -- $cfg
  for k in 400 400 20 20 20; do
  v=$(MI_SNARK_VS_LANES=$1 MI_SNARK_VS_SLOTS=$2 timeout -k 10 120 python bench.py --steps $k --warmup 5 --no-extra --no-cpu-baseline --acc-depth $1 --products $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f step %.4f steady %.4f' % (d['value'], d['ms_per_step'], d.get('steady_state_ms_per_step', 0)))")
  echo "lanes=$1 slots=$2 K=$k -> $v"
  done
done
