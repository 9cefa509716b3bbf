#!/bin/bash
# defaults against explicit lane / slot counts, interleaved, 20 steps and 400 steps
for rep in 1 2 3; do
 for cfg in "0 0" "12 8" "10 8" "10 6" "11 8"; do set -- $cfg
  A=""; [ $1 != 0 ] && A="--acc-depth $1 --products $2"
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 $A --no-extra --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lanes $1 slots $2 steps 20', round(d['value']), 'ms/step %.3f' % d['ms_per_step'])"
 done
done
for cfg in "12 8" "10 8" "10 6" "11 8"; do set -- $cfg
  timeout -k 10 200 python bench.py --steps 400 --warmup 3 --acc-depth $1 --products $2 --no-extra --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lanes $1 slots $2 steps 400', round(d['value']), 'ms/step %.3f' % d['ms_per_step'])"
done
rocm-smi --showuse 2>/dev/null | head -8
