#!/bin/bash
# the driver's shape (20 timed steps) and the long stream: when the stream driver starts sharing passes, and how many batches a pass takes
for cfg in "4 2" "1 2" "2 2" "1 4" "2 4" "1 3" "3 4"; do set -- $cfg
 for steps in 20 20 20 400; do
  MI_SNARK_STREAM_PAIR_AT=$1 MI_SNARK_STREAM_PER_PASS=$2 timeout -k 10 200 python bench.py --steps $steps --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pair_at $1 per_pass $2 steps $steps', round(d['value']), 'ms/step %.3f' % d['ms_per_step'])"
 done
done
