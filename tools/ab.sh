#!/bin/bash
# repeated short / long bench runs on the GPU box (run-to-run spread of the streamed headline)
for i in 1 2 3 4; do
python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(20, round(d['value']), round(d['ms_per_step'],4))"
done
for i in 1 2; do
python bench.py --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(400, round(d['value']), round(d['ms_per_step'],4), round(d.get('latency_ms_one_batch_unpipelined',0),3))"
done
