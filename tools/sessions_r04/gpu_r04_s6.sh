#!/bin/bash
# round 4, session 6: first GPU run of fbbev_msda_self_fused: module tests + S3 timing with / without + kernel stats
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward_projection.py -m gpu -q -s -x --timeout 600 -p no:cacheprovider > $OUT/r04_s6_pytest.log 2>&1; echo "pytest rc=$?"
grep -E "BackwardProjection full size|passed|failed|Error|error" $OUT/r04_s6_pytest.log | cut -c1-300 | tail -12
rm -f $OUT/r04_time_fb_msf.jsonl
for f in 1 0; do
  FBBEV_MSDA_FUSED=$f timeout 300 python tools/time_fb.py BL2 4 50 4 2>/dev/null >> $OUT/r04_time_fb_msf.jsonl
  FBBEV_MSDA_FUSED=$f timeout 300 python tools/time_fb.py REF 4 50 1 2>/dev/null >> $OUT/r04_time_fb_msf.jsonl
  FBBEV_MSDA_FUSED=$f timeout 300 python tools/time_fb.py REF 1 50 1 2>/dev/null >> $OUT/r04_time_fb_msf.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r04_time_fb_msf.jsonl'):
    d=json.loads(l); print(d['config'], d['B'], 'L', d['levels'], 'fb', round(d['ms_fb'],4), 'graph', round(d['ms_fb_graph'],4))
PY
cd /tmp; rm -rf $OUT/r04_prof_fb; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_prof_fb -- python $REPO/tools/time_fb.py BL2 4 20 4 > $OUT/r04_prof_fb.log 2>&1; cd $REPO
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r04_prof_fb/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
