# Poseidon-transcript stream against the LDS pad (block exclusivity) -- for builds with other transcript block sizes
cd $GRAFT_REPO_ROOT
for pad in 20000 40000 1000; do
  MI_SNARK_FS_LDS_PAD=$pad timeout -k 10 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extra']['batch_1024_poseidon_fs']
print('pad=$pad  poseidon streamed %.0f (%.3f ms/batch)  latency %.2f  kernel %.2f ms' % (e['streamed_proofs_per_s'], e['streamed_ms_per_batch'], e['latency_ms_one_batch_unpipelined'], e['transcript_kernel_ms']))"
done
