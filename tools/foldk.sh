#!/bin/bash
# Run on the GPU box: the 20-batch and 400-batch rates and the one-batch latency for the fold's split factors.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for cfg in "2 4" "1 4" "2 6" "1 6" "3 4" "2 4"; do  # (the knobs existed for this sweep only: msm.hip now has 3 / 4 fixed)
  set -- $cfg
  for rep in 1 2; do
  MI_SNARK_FOLD_K_BUSY=$1 MI_SNARK_FOLD_K_ALONE=$2 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('busy $1 alone $2: 20 steps %.3e  one batch %.4f ms' % (d['value'], d['latency_ms_one_batch_unpipelined']))"
  done
  MI_SNARK_FOLD_K_BUSY=$1 MI_SNARK_FOLD_K_ALONE=$2 python bench.py --steps 400 --warmup 20 --no-extra --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('busy $1 alone $2: 400 steps %.3e' % (d['value']))"
done
