#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_backward_projection.py -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_bp.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_bp.log | cut -c1-300
rm -f $OUT/r03_time_fb_rows_linear.jsonl
for v in x3 f32; do
  FBBEV_ROWS_LINEAR=$v timeout 300 python tools/time_fb.py BL2 4 30 4 2>/dev/null | sed "s/^{/{\"rows_linear\": \"$v\", /" >> $OUT/r03_time_fb_rows_linear.jsonl
  FBBEV_ROWS_LINEAR=$v timeout 300 python tools/time_fb.py REF 4 30 1 2>/dev/null | sed "s/^{/{\"rows_linear\": \"$v\", /" >> $OUT/r03_time_fb_rows_linear.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r03_time_fb_rows_linear.jsonl'):
    d=json.loads(l); print(d['rows_linear'], d['config'], d['B'], 'fb', round(d['ms_fb'],4), 'graph', d.get('ms_fb_graph'))
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fb_x3 -- python $REPO/tools/time_fb.py BL2 4 20 4 > $OUT/prof_fb_x3.log 2>&1
cd $REPO
python - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/prof_fb_x3/**/*kernel_stats.csv',recursive=True))[-1]
rows=list(csv.DictReader(open(f)))
calls=max(int(r['Calls']) for r in rows if 'k_da_cross_attn_fwd' in r['Name'])
for r in rows[:16]: print(r['Name'][:90].ljust(90), r['Calls'], round(float(r['TotalDurationNs'])/calls/1e3,1),'us/iter')
PY
find $OUT -name "*kernel_trace.csv" -size +3M -delete
