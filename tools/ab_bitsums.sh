# A/B of the small-MSM bit-sum reduction: block tree (default) against one wave per group (MI_SNARK_BITSUM_LPG=1)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lpg in 0 1; do
  v=$(MI_SNARK_BITSUM_LPG=$lpg timeout -k 10 120 python bench.py --steps 400 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f step %.4f steady %.4f latency %.3f bitsums %s' % (d['value'], d['ms_per_step'], d.get('steady_state_ms_per_step', 0), d['latency_ms_one_batch_unpipelined'], d['kernels'].get('msm_bitsums')))")
  echo "lpg=$lpg 400 steps -> $v"
  v=$(MI_SNARK_BITSUM_LPG=$lpg timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f step %.4f' % (d['value'], d['ms_per_step']))")
  echo "lpg=$lpg  20 steps -> $v"
done
done
