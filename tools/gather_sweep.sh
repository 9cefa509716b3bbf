#!/bin/bash
# Run on the GPU box: the gather stage's length (MI_SNARK_MSM_GATHER: 0 off, 1 on with the default 8, n > 1: n chunk partials
# per entry -- forced on for every small MSM) against the accumulation's chunk length (MI_SNARK_MSM_SMALL_T).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
IFS=, read -ra LIST <<< "${CFGS:-8 16,4 16,16 16,8 8,16 8,8 12,12 12,8 24,8 16}"
for cfg in "${LIST[@]}"; do
  set -- $cfg
  for steps in 400 20 20; do
  MI_SNARK_MSM_GATHER=$1 MI_SNARK_MSM_SMALL_T=$2 timeout -k 10 200 python bench.py --steps $steps --warmup $((steps / 20 + 4)) --no-extra --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('gather $1 chunk $2  %3d steps %.3e  one batch %.3f' % ($steps, d['value'], d['latency_ms_one_batch_unpipelined']))"
  done
done
