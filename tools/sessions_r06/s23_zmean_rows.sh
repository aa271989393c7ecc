#!/bin/bash
# round 6, session 23: the Z-mean written as query rows + bev_embedding (no transposing pass in front of the encoder): same-box A/B
REPO=$(pwd); OUT=$REPO/gpurun_out/s23; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/time_fb.jsonl
for rep in 1 2 3; do
for zr in 0 1; do
  for cfg in "BL2 4 40 4" "REF 4 40 1" "BL2 1 40 4"; do
    FBBEV_ZMEAN_ROWS=$zr timeout 300 python tools/time_fb.py $cfg 2>/dev/null | sed "s/^{/{\"zmean_rows\": $zr, /" >> $OUT/time_fb.jsonl
  done
done
done
python - <<'PY'
import json
for l in open('gpurun_out/s23/time_fb.jsonl'):
    d = json.loads(l); print('zmean_rows', d['zmean_rows'], d['config'], d['B'], d['levels'], 'eager', round(d['ms_fb'], 4), 'graph', round(d.get('ms_fb_graph') or 0, 4))
PY
timeout 1500 python -m pytest tests/test_gpu_backward_projection.py tests/test_gpu_full_model.py tests/test_gpu_parity.py -q -x -p no:cacheprovider 2>&1 | tail -2
