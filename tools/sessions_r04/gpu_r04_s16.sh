#!/bin/bash
# round 4, session 16: the last len % 4 points of an interval as one masked gather batch: dense pool kernel per launch, all shape classes + S2 / S3
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/r04_pool_tail_batch.jsonl
for cfg in "REF 16 f32" "REF 4 f32" "REF 1 f32" "BL2 16 f32" "BL2 16 bf16" "BL5 1 f32" "BL1 1 f32"; do
  timeout 200 python tools/time_pool_flags.py $cfg 2>/dev/null | head -1 >> $OUT/r04_pool_tail_batch.jsonl
  timeout 200 python tools/time_pool_flags.py $cfg 2>/dev/null | head -1 >> $OUT/r04_pool_tail_batch.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r04_pool_tail_batch.jsonl'):
    d=json.loads(l); print(d['config'], d['B'], d.get('storage'), d.get('tv'), d.get('flags'), d.get('ms'), d.get('frac_of_8TBs'))
PY
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "pool or dense or fused_module or full_size or 16bit" 2>&1 | tail -3
timeout 300 python tools/time_fb.py REF 4 50 1 2>/dev/null | cut -c1-400
timeout 300 python tools/time_fb.py BL2 4 50 4 2>/dev/null | cut -c1-400
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt-storage 2>/dev/null | cut -c1-330
