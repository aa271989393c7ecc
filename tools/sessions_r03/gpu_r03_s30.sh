#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_backward_projection.py tests/test_gpu_full_model.py -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_bp.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_bp.log | cut -c1-300
rm -f $OUT/r03_time_fb_final.jsonl
for i in 1 2; do
  timeout 300 python tools/time_fb.py BL2 4 50 4 2>/dev/null >> $OUT/r03_time_fb_final.jsonl
  timeout 300 python tools/time_fb.py REF 4 50 1 2>/dev/null >> $OUT/r03_time_fb_final.jsonl
done
timeout 300 python tools/time_fb.py REF 1 50 1 2>/dev/null >> $OUT/r03_time_fb_final.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r03_time_fb_final.jsonl'):
    d=json.loads(l); print(d['config'], d['B'], 'fb', round(d['ms_fb'],4), 'graph', d.get('ms_fb_graph'))
PY
