#!/bin/bash
# one rank with the exchange path switched on (communicator of world 1): cost of the grouped exchange in the streaming driver
for x in host rccl; do for g in 1 4; do
  MI_SNARK_EXCHANGE=$([ $x = host ] && echo host || echo auto) timeout -k 10 200 python bench.py --steps ${1:-200} --warmup 5 --exchange native --exchange-group $g --no-extra --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$x group $g', round(d['value']), 'ms/step %.3f steady %.3f' % (d['ms_per_step'], d['steady_state_ms_per_step']), c['exchange'], c['exchange_calls'], c['exchange_wait_ms_per_group'])"
done; done
timeout -k 10 200 python bench.py --steps ${1:-200} --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('no exchange', round(d['value']), 'ms/step %.3f steady %.3f' % (d['ms_per_step'], d['steady_state_ms_per_step']))"
