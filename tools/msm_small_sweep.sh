#!/bin/bash
# Run on the GPU box: the small-MSM plan's knobs (chunk length, window bias) under merged passes of up to four batches.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
IFS=, read -ra LIST <<< "${CFGS:-16 -1,8 -1,12 -1,24 -1,32 -1,16 -2,16 0,24 0,16 -1}"
for cfg in "${LIST[@]}"; do
  set -- $cfg
  for steps in 400 20 20; do
  MI_SNARK_MSM_SMALL_T=$1 MI_SNARK_MSM_SMALL_C_BIAS=$2 timeout -k 10 200 python bench.py --steps $steps --warmup $((steps / 20 + 4)) --no-extra --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('chunk $1 c bias $2  %3d steps %.3e  one batch %.3f' % ($steps, d['value'], d['latency_ms_one_batch_unpipelined']))"
  done
done
