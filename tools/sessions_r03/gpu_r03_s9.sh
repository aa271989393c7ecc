#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/r03_time_rank_seg.jsonl
for cfg in "BL2 16" "REF 16" "BL2 4" "REF 1" "BL5 4" "BL1 1"; do
  for seg in 0 1; do
    FBBEV_RANK_SEG=$([ $seg = 0 ] && echo 0 || echo -1) timeout 120 python tools/time_rank.py $cfg 2>/dev/null | sed "s/^{/{\"segmented\": $seg, /" >> $OUT/r03_time_rank_seg.jsonl
  done
done
cut -c1-210 $OUT/r03_time_rank_seg.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_parity.log 2>&1; echo "pytest parity rc=$?"; tail -3 $OUT/pytest_parity.log | cut -c1-200
timeout 600 python bench.py --steps 30 --warmup 5 > $OUT/bench_seg.json 2> $OUT/bench_seg.err; echo "bench rc=$?"; cut -c1-330 $OUT/bench_seg.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_seg -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-storage > $OUT/prof_stats_seg.log 2>&1; echo "rocprof rc=$?"
cd $REPO
find $OUT -name "*kernel_trace.csv" -size +3M -delete
