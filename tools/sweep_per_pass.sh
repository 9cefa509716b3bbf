# Poseidon-transcript and v2-shape streams against batches per pass
cd $GRAFT_REPO_ROOT
for pp in 2 3 4; do
  MI_SNARK_STREAM_PER_PASS=$pp timeout -k 10 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extra']
print('per_pass=$pp headline %.0f  poseidon %.0f (%.3f ms/batch)  v2 %.0f' % (d['value'], e['batch_1024_poseidon_fs']['streamed_proofs_per_s'], e['batch_1024_poseidon_fs']['streamed_ms_per_batch'], e['batch_1024_v2_shape']['streamed_proofs_per_s']))"
done
