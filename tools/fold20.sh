#!/bin/bash
# the driver's shape (20 timed steps): host fold threads
for ft in 3 6 10; do for rep in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --fold-threads $ft --no-extra --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fold-threads $ft', round(d['value']), 'ms/step %.3f' % d['ms_per_step'])"
done; done
