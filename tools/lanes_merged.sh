#!/bin/bash
# Run on the GPU box: lane / product-slot counts with merged passes (a pass ends in ONE product: fewer slots are needed).
# MI_SNARK_VS_LANES / _SLOTS set what the device creates, --acc-depth / --products what the stream uses.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
 for cfg in "12 8" "14 6" "14 4" "12 4" "10 4" "8 4" "16 4"; do set -- $cfg
  for steps in 400 20 20; do
  MI_SNARK_VS_LANES=$1 MI_SNARK_VS_SLOTS=$2 timeout -k 10 200 python bench.py --steps $steps --warmup $((steps / 20 + 4)) --acc-depth $1 --products $2 --no-extra --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lanes $1 slots $2  %3d steps %.3e' % ($steps, d['value']))"
  done
 done
done
