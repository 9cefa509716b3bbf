# streamed throughput against the number of host fold threads
cd $GRAFT_REPO_ROOT
for ft in 3 4 5 6 8 4 6; do
  v=$(timeout -k 10 120 python bench.py --steps 400 --warmup 5 --no-extra --no-cpu-baseline --fold-threads $ft 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f step %.4f steady %.4f latency %.3f' % (d['value'], d['ms_per_step'], d.get('steady_state_ms_per_step', 0), d['latency_ms_one_batch_unpipelined']))")
  echo "fold_threads=$ft -> $v"
done
