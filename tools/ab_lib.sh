#!/bin/bash
# Run on the GPU box: A/B of two builds of the library (MI_SNARK_LIB selects the other one), streamed and short rates.
#   tools/ab_lib.sh build/libother.so [repetitions]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OTHER=$R/$1
REPS=${2:-3}
for rep in $(seq $REPS); do
for lib in "" "$OTHER"; do
  for steps in 400 20; do
  MI_SNARK_LIB=$lib python bench.py --steps $steps --warmup $((steps / 20 + 4)) --no-extra --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-28s %3d steps %.3e' % ('${lib:-(this build)}'[-28:], $steps, d['value']))"
  done
done
done
