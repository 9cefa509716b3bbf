#!/bin/bash
# the driver's shape (20 timed steps), one rank, exchange path on: lanes/slots/group variants (three runs each)
for cfg in "auto 4 8 6" "auto 4 10 6" "auto 2 10 6" "auto 1 10 6" "host 4 12 8" "host 2 12 8" "none 1 12 8"; do set -- $cfg
  for rep in 1 2 3; do
  X="--exchange native --exchange-group $2"; [ $1 = none ] && X=""
  MI_SNARK_EXCHANGE=$1 timeout -k 10 200 python bench.py --steps 20 --warmup 3 $X --acc-depth $3 --products $4 --no-extra --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$cfg', round(d['value']), 'ms/step %.3f' % d['ms_per_step'], c['exchange'])"
  done
done
