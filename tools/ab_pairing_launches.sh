cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do
for v in 1 2; do
  MI_SNARK_PAIRING_LAUNCHES=$v timeout -k 10 120 python bench.py --steps 400 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('launches=$v', round(d['value']), round(d['ms_per_step'],4), round(d['steady_state_ms_per_step'],4), round(d['latency_ms_one_batch_unpipelined'],3), 'pairing_us', round(d['roofline']['avg_launch_us']))"
done; done
