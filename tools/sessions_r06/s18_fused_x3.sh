#!/bin/bash
# round 6, session 18: the one-kernel history step at fp32 grade (k_history_fused_x3): bits vs the two kernels, time
REPO=$(pwd); OUT=$REPO/gpurun_out/s18; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tests/_tmp_gpu_fused_x3.py 2>&1 | tail -3
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], "fused_x3", d["fused_x3"], "step_ms", d["fused_ms"], "warp_ms", d["warp_ms"], "err", d["bf16_convs_vs_fp32_convs_max_rel_to_peak"])'
run() { timeout 600 python tools/time_history.py 400 400 16 1 $2 noref cx3 vm 2>>$OUT/err1.log | tee -a $OUT/hist.jsonl | python -c "$P" "$1"; }
for rep in 1 2; do
  HIST_FUSED_X3=0 run "two kernels f16" f16
  HIST_FUSED_X3=1 run "one kernel f16" f16
done
cd /tmp && HIST_FUSED_X3=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $REPO/tools/time_history.py 400 400 16 1 f16 noref cx3 vm > $OUT/prof.log 2>&1
cd $REPO
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); head -5 "$f" | cut -c1-200
