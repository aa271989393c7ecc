#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/r03_time_fb_fold.jsonl
for i in 1 2; do
 for f in 1 0; do
  FBBEV_ROWS_LINEAR_FOLD=$f timeout 300 python tools/time_fb.py BL2 4 50 4 2>/dev/null | sed "s/^{/{\"fold\": $f, /" >> $OUT/r03_time_fb_fold.jsonl
  FBBEV_ROWS_LINEAR_FOLD=$f timeout 300 python tools/time_fb.py REF 4 50 1 2>/dev/null | sed "s/^{/{\"fold\": $f, /" >> $OUT/r03_time_fb_fold.jsonl
 done
done
python - <<'PY'
import json
for l in open('gpurun_out/r03_time_fb_fold.jsonl'):
    d=json.loads(l); print('fold', d['fold'], d['config'], d['B'], 'fb', round(d['ms_fb'],4), 'graph', round(d.get('ms_fb_graph') or 0,4))
PY
