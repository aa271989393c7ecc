#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_history.py -m gpu -q --timeout 900 -p no:cacheprovider -k "split_operand or fused_warp" > $OUT/pytest_hist.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_hist.log | cut -c1-400
rm -f $OUT/r03_time_history_x3.jsonl
for c in "" cbf16 cx3; do
  timeout 300 python tools/time_history.py 400 400 16 1 f16 noref $c vm unfused 2>/dev/null >> $OUT/r03_time_history_x3.jsonl
done
timeout 300 python tools/time_history.py 100 100 8 1 f16 noref cx3 vm unfused 2>/dev/null >> $OUT/r03_time_history_x3.jsonl
timeout 300 python tools/time_history.py 100 100 8 1 f16 noref vm unfused 2>/dev/null >> $OUT/r03_time_history_x3.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r03_time_history_x3.jsonl'):
    d=json.loads(l); print(d['grid'], d['history_dtype'], d['conv_compute'], 'step ms', d['fused_ms'], 'warp', d['warp_ms'], 'rel to fp32', d['bf16_convs_vs_fp32_convs_max_rel_to_peak'])
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_hist_x3 -- python $REPO/tools/time_history.py 400 400 16 1 f16 noref cx3 vm unfused > $OUT/prof_hist_x3.log 2>&1
cd $REPO
python - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/prof_hist_x3/**/*kernel_stats.csv',recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:6]: print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
find $OUT -name "*kernel_trace.csv" -size +3M -delete
