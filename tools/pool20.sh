#!/bin/bash
# the driver's shape (20 timed steps): host worker pool size x fold threads
for cfg in "8 3" "12 3" "12 6" "16 6" "24 8"; do set -- $cfg
 for rep in 1 2 3; do
  MI_SNARK_POOL=$1 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --fold-threads $2 --no-extra --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pool $1 fold-threads $2', round(d['value']), 'ms/step %.3f' % d['ms_per_step'])"
 done
done
nproc; python -c "import os; print(len(os.sched_getaffinity(0)))"
