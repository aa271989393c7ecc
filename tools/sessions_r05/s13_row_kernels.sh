#!/bin/bash
# round 5, session 13: register-held weight staging + batched epilogue loads of the row kernels (k_rows_ffn_x3, k_rows_linear_x3):
# parity tests of the block kernels + backward projection, S3 timing, kernel trace
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TAG=${1:-s13}
timeout 1500 python -m pytest tests/test_gpu_block_kernels.py tests/test_gpu_backward_projection.py tests/test_gpu_full_model.py -m gpu -q -s --timeout 900 -p no:cacheprovider > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"
tail -2 $OUT/${TAG}_pytest.log | cut -c1-300
grep -E "^E  |FAILED" $OUT/${TAG}_pytest.log | head
grep -o "\[observed\].*" $OUT/${TAG}_pytest.log | grep -i "ffn\|rows_linear" | head -12
rm -f $OUT/${TAG}_time_fb.jsonl
for rep in 1 2; do
  for cfg in "BL2 4 40 4" "REF 1 40 1" "REF 4 40 1"; do
    timeout 300 python tools/time_fb.py $cfg 2>/dev/null | sed "s/^{/{\"knobs\": \"\", /" >> $OUT/${TAG}_time_fb.jsonl
  done
done
python - $TAG <<'PY'
import json, sys
for l in open('gpurun_out/%s_time_fb.jsonl' % sys.argv[1]):
    d = json.loads(l); print(d['knobs'] or 'default', d['config'], d['B'], d['levels'], 'eager', round(d['ms_fb'], 4), 'graph', round(d['ms_fb_graph'], 4))
PY
rm -rf $OUT/${TAG}_prof
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -- python $REPO/tools/time_fb.py BL2 4 20 4 > $OUT/${TAG}_prof.log 2>&1; echo "rocprof rc=$?"
cd $REPO
python - $TAG <<'PY'
import csv, glob, sys
for f in glob.glob('gpurun_out/%s_prof/**/*kernel_stats.csv' % sys.argv[1], recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(r['Name'][:90], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), r['Percentage'])
PY
