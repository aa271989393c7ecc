#!/bin/bash
# round 6, session 22: the last encoder layer's tail + FFN kernel writes the refined BEV as (B, C, Y, X) itself (no transposing pass): same-box A/B
REPO=$(pwd); OUT=$REPO/gpurun_out/s22; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/time_fb.jsonl
for rep in 1 2 3; do
for pl in 0 1; do
  for cfg in "BL2 4 40 4" "REF 1 40 1" "REF 4 40 1"; do
    FBBEV_BP_OUT_PLANES=$pl timeout 300 python tools/time_fb.py $cfg 2>/dev/null | sed "s/^{/{\"out_planes\": $pl, /" >> $OUT/time_fb.jsonl
  done
done
done
python - <<'PY'
import json
for l in open('gpurun_out/s22/time_fb.jsonl'):
    d = json.loads(l); print('out_planes', d['out_planes'], d['config'], d['B'], d['levels'], 'eager', round(d['ms_fb'], 4), 'graph', round(d.get('ms_fb_graph') or 0, 4))
PY
timeout 900 python -m pytest tests/test_gpu_backward_projection.py tests/test_gpu_fb_view_transform.py -q -x -p no:cacheprovider 2>&1 | tail -2
