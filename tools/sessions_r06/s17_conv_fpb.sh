#!/bin/bash
# round 6, session 17: split-operand convolutions, two frames per workgroup barrier (four W2 buffers) vs one
REPO=$(pwd); OUT=$REPO/gpurun_out/s17; mkdir -p $OUT; export TMPDIR=/tmp
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], "fused_ms", d["fused_ms"], "warp_ms", d["warp_ms"], "conv+frame_ms", round(d["fused_ms"]-d["warp_ms"],3), "err", d["bf16_convs_vs_fp32_convs_max_rel_to_peak"])'
run() { HIST_PIPE=0 timeout 600 python tools/time_history.py 400 400 16 1 $2 noref cx3 vm 2>>$OUT/err1.log | tee -a $OUT/hist.jsonl | python -c "$P" "$1"; }
for rep in 1 2 3; do
  FBBEV_HX3_FPB=1 run "fpb=1 f16" f16
  FBBEV_HX3_FPB=2 run "fpb=2 f16" f16
done
FBBEV_HX3_FPB=1 run "fpb=1 bf16" bf16
FBBEV_HX3_FPB=2 run "fpb=2 bf16" bf16
FBBEV_HX3_FPB=2 timeout 900 python -m pytest tests/test_gpu_history.py -q -x -p no:cacheprovider -k "split_operand or config4 or fixture" 2>&1 | tail -2
