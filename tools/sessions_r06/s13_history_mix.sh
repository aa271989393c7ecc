#!/bin/bash
# round 6, session 13: warp with the half widened by the multiply (v_fma_mix_f32), conv-1 bias as the accumulator's start
REPO=$(pwd); OUT=$REPO/gpurun_out/s13; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_history.py -q -x -p no:cacheprovider 2>&1 | tail -3
for i in 1 2; do
  timeout 600 python tools/time_history.py 400 400 16 1 f16 noref cx3 vm 2>>$OUT/err1.log | tee -a $OUT/hist.jsonl | cut -c1-330
  FBBEV_HISTORY_VM_TU=4 timeout 600 python tools/time_history.py 400 400 16 1 f16 noref cx3 vm 2>>$OUT/err1.log | sed 's/^{/{"TU": 4, /' | tee -a $OUT/hist.jsonl | cut -c1-330
done
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $REPO/tools/time_history.py 400 400 16 1 f16 noref cx3 vm > $OUT/prof.log 2>&1
cd $REPO
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); head -5 "$f" | cut -c1-200
