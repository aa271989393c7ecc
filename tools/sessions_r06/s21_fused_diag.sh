#!/bin/bash
# (the FBBEV_HFX_DIAG builds this script timed were removed again after the session: results in profiles/r06_exp_history_step.md)
# round 6, session 21: where the one-kernel history step spends its time -- diagnostic builds that leave one part out (wrong results)
REPO=$(pwd); OUT=$REPO/gpurun_out/s21; mkdir -p $OUT; export TMPDIR=/tmp
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], "step_ms", d["fused_ms"], "warp_ms", d["warp_ms"])'
run() { HIST_FUSED_X3=1 timeout 600 python tools/time_history.py 400 400 16 1 f16 noref cx3 vm 2>>$OUT/err1.log | tee -a $OUT/hist.jsonl | python -c "$P" "$1"; }
for rep in 1 2; do
FBBEV_HFX_DIAG=0 run "full"
FBBEV_HFX_DIAG=1 run "no W2 DMA"
FBBEV_HFX_DIAG=2 run "no tap requests"
FBBEV_HFX_DIAG=3 run "no ring stores"
done
