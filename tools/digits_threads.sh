#!/bin/bash
# Run on the GPU box: streamed rates for the digits kernel's workgroup sizes (experiment knob MI_SNARK_DIGITS_THREADS).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
for t in 1024 512 256; do  # (the knob existed for this sweep only: msm.hip launches 512)
  MI_SNARK_DIGITS_THREADS=$t python bench.py --steps 400 --warmup 20 --no-extra --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('threads $t: 400 steps %.3e' % (d['value']))"
  MI_SNARK_DIGITS_THREADS=$t python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('threads $t: 20 steps %.3e one batch %.4f' % (d['value'], d['latency_ms_one_batch_unpipelined']))"
done
done
