#!/bin/bash
# Run on the GPU box: streamed and short rates with the passes' batches merged into one aggregate (default) and kept apart.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do
for m in 1 0; do
  for steps in 400 20; do
  MI_SNARK_STREAM_MERGE=$m python bench.py --steps $steps --warmup $((steps / 20 + 4)) --no-extra --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('merge $m  %3d steps %.3e' % ($steps, d['value']))"
  done
done
done
