#!/bin/bash
# the driver's shape (20 timed steps): accumulation lanes x product slots
for cfg in "4 8" "6 8" "8 8" "10 8" "12 8" "6 12" "8 12"; do set -- $cfg
 for rep in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --acc-depth $1 --products $2 --no-extra --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lanes $1 slots $2', round(d['value']), 'ms/step %.3f' % d['ms_per_step'])"
 done
done
