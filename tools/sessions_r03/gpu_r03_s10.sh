#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/r03_time_rank_seg2.jsonl
for cfg in "BL2 16" "REF 16" "BL2 4" "BL5 4"; do
  for seg in 0 2; do
    FBBEV_RANK_SEG=$seg timeout 120 python tools/time_rank.py $cfg 2>/dev/null | sed "s/^{/{\"seg_mode\": $seg, /" >> $OUT/r03_time_rank_seg2.jsonl
  done
done
cut -c1-140 $OUT/r03_time_rank_seg2.jsonl
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_seg2 -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-storage > $OUT/prof_stats_seg2.log 2>&1; echo "rocprof rc=$?"
cd $REPO
python - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/prof_stats_seg2/**/*kernel_stats.csv',recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:14]: print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
find $OUT -name "*kernel_trace.csv" -size +3M -delete
