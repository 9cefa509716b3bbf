#!/bin/bash
# streamed verifier throughput against the small-MSM chunk length and window width (run on the GPU box)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "16 -1" "12 -1" "20 -1" "24 -1" "16 0" "16 -2" "12 0" "20 0"; do
  set -- $cfg
  v=$(MI_SNARK_MSM_SMALL_T=$1 MI_SNARK_MSM_SMALL_C_BIAS=$2 timeout -k 10 120 python bench.py --steps 400 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f %.4f %.4f lat %.3f' % (d['value'], d['ms_per_step'], d.get('steady_state_ms_per_step', 0), d.get('latency_ms_one_batch_unpipelined', 0)))")
  echo "T=$1 c_bias=$2 -> $v"
done
