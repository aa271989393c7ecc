#!/bin/bash
# round 4, session 12: history warp with v_fma_mix_f32 in the fp16 blend; history GPU tests; configs[4] step with the new default
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_history.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -2
rm -f $OUT/r04_time_history.jsonl
for i in 1 2; do
  timeout 300 python tools/time_history.py 400 400 16 1 f16 noref vm cx3 >> $OUT/r04_time_history.jsonl 2>/dev/null
  timeout 300 python tools/time_history.py 100 100 8 4 f16 noref vm cx3 >> $OUT/r04_time_history.jsonl 2>/dev/null
done
python - <<'PY'
import json
for l in open('gpurun_out/r04_time_history.jsonl'):
    d=json.loads(l); print(d['grid'], d['B'], d['history_dtype'], d['conv_compute'], 'fused_ms', d['fused_ms'], 'warp_ms', d['warp_ms'], d['warp_GBps_read_plus_write'])
PY
