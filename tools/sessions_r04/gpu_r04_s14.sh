#!/bin/bash
# round 4, session 14: k_rows_linear_x3 with 16 rows per wave at 3 waves / SIMD (FBBEV_ROWS_LINEAR_NT=1) vs 32 rows at 2
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/r04_time_fb_nt.jsonl
for nt in 1 2 1 2; do
  FBBEV_ROWS_LINEAR_NT=$nt timeout 300 python tools/time_fb.py BL2 4 50 4 2>/dev/null | sed "s/^{/{\"rows_nt\": $nt, /" >> $OUT/r04_time_fb_nt.jsonl
  FBBEV_ROWS_LINEAR_NT=$nt timeout 300 python tools/time_fb.py REF 4 50 1 2>/dev/null | sed "s/^{/{\"rows_nt\": $nt, /" >> $OUT/r04_time_fb_nt.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r04_time_fb_nt.jsonl'):
    d=json.loads(l); print('rows_nt', d['rows_nt'], d['config'], d['B'], 'fb', round(d['ms_fb'],4), 'graph', round(d['ms_fb_graph'],4))
PY
