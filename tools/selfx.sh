#!/bin/bash
# one rank with the exchange path switched on (a communicator of world 1): what the grouped exchange of the streaming
# driver costs by itself, per transport (auto = RCCL, host = shared memory), exchange group, lanes and product slots
for cfg in "auto 4 8 6" "auto 4 10 6" "auto 8 8 6" "host 4 8 6" "host 4 12 8"; do set -- $cfg
  MI_SNARK_EXCHANGE=$1 timeout -k 10 200 python bench.py --steps 200 --warmup 5 --exchange native --exchange-group $2 --acc-depth $3 --products $4 --no-extra --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$cfg', round(d['value']), 'ms/step %.3f steady %.3f' % (d['ms_per_step'], d['steady_state_ms_per_step']), c['exchange'], c['exchange_calls'], c['exchange_wait_ms_per_group'])"
done
