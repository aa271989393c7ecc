#!/bin/bash
# round 4, session 17: the FFN pair as one kernel (fbbev_rows_ffn_x3): tests + S3 on / off + kernel stats
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward_projection.py tests/test_gpu_full_model.py -m gpu -q -s -x --timeout 600 -p no:cacheprovider > $OUT/r04_s17_pytest.log 2>&1; echo "pytest rc=$?"
grep -E "BackwardProjection full size|passed|failed|Error|error" $OUT/r04_s17_pytest.log | cut -c1-300 | tail -5
rm -f $OUT/r04_time_fb_ffn.jsonl
for f in 1 0 1 0; do
  FBBEV_FUSE_FFN=$f timeout 300 python tools/time_fb.py BL2 4 50 4 2>/dev/null | sed "s/^{/{\"fuse_ffn\": $f, /" >> $OUT/r04_time_fb_ffn.jsonl
  FBBEV_FUSE_FFN=$f timeout 300 python tools/time_fb.py REF 4 50 1 2>/dev/null | sed "s/^{/{\"fuse_ffn\": $f, /" >> $OUT/r04_time_fb_ffn.jsonl
  FBBEV_FUSE_FFN=$f timeout 300 python tools/time_fb.py REF 1 50 1 2>/dev/null | sed "s/^{/{\"fuse_ffn\": $f, /" >> $OUT/r04_time_fb_ffn.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r04_time_fb_ffn.jsonl'):
    d=json.loads(l); print('fuse_ffn', d['fuse_ffn'], d['config'], d['B'], 'fb', round(d['ms_fb'],4), 'graph', round(d['ms_fb_graph'],4))
PY
cd /tmp; rm -rf $OUT/r04_prof_fb; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_prof_fb -- python $REPO/tools/time_fb.py BL2 4 20 4 > $OUT/r04_prof_fb.log 2>&1; cd $REPO
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r04_prof_fb/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
