#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_history.py -m gpu -q -k fused --timeout 900 -p no:cacheprovider > $OUT/pytest_hist.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_hist.log | cut -c1-300
rm -f $OUT/r03_time_history_fused_v4.jsonl
for dt in f16 bf16; do
  timeout 300 python tools/time_history.py 400 400 16 1 $dt noref cbf16 vm unfused 2>/dev/null >> $OUT/r03_time_history_fused_v4.jsonl
  timeout 300 python tools/time_history.py 400 400 16 1 $dt noref cbf16 vm 2>/dev/null >> $OUT/r03_time_history_fused_v4.jsonl
done
timeout 300 python tools/time_history.py 100 100 8 1 f16 noref cbf16 vm unfused 2>/dev/null >> $OUT/r03_time_history_fused_v4.jsonl
timeout 300 python tools/time_history.py 100 100 8 1 f16 noref cbf16 vm 2>/dev/null >> $OUT/r03_time_history_fused_v4.jsonl
timeout 300 python tools/time_history.py 200 200 16 4 f16 noref cbf16 vm unfused 2>/dev/null >> $OUT/r03_time_history_fused_v4.jsonl
timeout 300 python tools/time_history.py 200 200 16 4 f16 noref cbf16 vm 2>/dev/null >> $OUT/r03_time_history_fused_v4.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r03_time_history_fused_v4.jsonl'):
    d=json.loads(l); print(d['grid'], d['B'], d['history_dtype'], 'one_kernel', d['warp_conv_one_kernel'], 'step ms', d['fused_ms'], 'warp alone', d['warp_ms'])
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_hist_fused4 -- python $REPO/tools/time_history.py 400 400 16 1 f16 noref cbf16 vm > $OUT/prof_hist_fused4.log 2>&1
cd $REPO
python - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/prof_hist_fused4/**/*kernel_stats.csv',recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:8]: print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
find $OUT -name "*kernel_trace.csv" -size +3M -delete
