#!/bin/bash
# Run on the GPU box: the small MSM's gather stage (shorter chunk lists) ahead of the per-bit sums (MI_SNARK_MSM_GATHER=0/1), streamed,
# short and one batch alone.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do
for m in 1 0; do
  for steps in 400 20 20; do
  MI_SNARK_MSM_GATHER=$m python bench.py --steps $steps --warmup $((steps / 20 + 4)) --no-extra --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('gather $m  %3d steps %.3e  one batch %.3f ms' % ($steps, d['value'], d['latency_ms_one_batch_unpipelined']))"
  done
done
done
