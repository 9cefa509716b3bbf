# A/B of the bit-sum reduction of multi-batch passes: groups of 64 lanes (one per wave, default) or 32 lanes; medians
cd $GRAFT_REPO_ROOT
for part in 0 32 0 32 0 32 0 32 0 32; do
  v=$(MI_SNARK_BITSUM_PART=$part timeout -k 10 120 python bench.py --steps 400 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f %.4f %.4f' % (d['value'], d['ms_per_step'], d.get('steady_state_ms_per_step', 0)))")
  echo "part=$part $v"
done
