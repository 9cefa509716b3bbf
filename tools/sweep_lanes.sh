# streamed throughput against batches per pass, lanes and product slots (lanes + slots = 20 HIP streams)
cd $GRAFT_REPO_ROOT
for cfg in "2 10 10 4" "4 10 10 4" "4 8 12 4" "4 6 14 4" "3 8 12 4" "4 8 12 2" "4 7 13 4" "3 10 10 4"; do
  set -- $cfg
  for k in 400 400 20; do
  v=$(MI_SNARK_STREAM_PER_PASS=$1 MI_SNARK_STREAM_PAIR_AT=$4 MI_SNARK_VS_LANES=$2 MI_SNARK_VS_SLOTS=$3 timeout -k 10 120 python bench.py --steps $k --warmup 5 --no-extra --no-cpu-baseline --acc-depth $2 --products $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f step %.4f steady %.4f latency %.3f' % (d['value'], d['ms_per_step'], d.get('steady_state_ms_per_step', 0), d['latency_ms_one_batch_unpipelined']))")
  echo "per_pass=$1 lanes=$2 slots=$3 pair_at=$4 K=$k -> $v"
  done
done
