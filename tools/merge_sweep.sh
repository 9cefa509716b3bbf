#!/bin/bash
# Run on the GPU box: merged passes -- how many batches a pass takes and from how many open passes on the driver shares.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
IFS=, read -ra LIST <<< "${CFGS:-2 4,3 4,4 4,3 2,4 2,4 6,2 4}"
for cfg in "${LIST[@]}"; do
  set -- $cfg
  for steps in 400 20 20; do
  MI_SNARK_STREAM_PER_PASS=$1 MI_SNARK_STREAM_PAIR_AT=$2 python bench.py --steps $steps --warmup $((steps / 20 + 4)) --no-extra --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('per pass $1 share at $2  %3d steps %.3e' % ($steps, d['value']))"
  done
done
