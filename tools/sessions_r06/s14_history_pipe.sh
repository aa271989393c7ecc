#!/bin/bash
# round 6, session 14: the history step as a two-stream pipeline over bands of rows (fbbev_history_step_x3_vm): chunk-count sweep
REPO=$(pwd); OUT=$REPO/gpurun_out/s14; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_history.py -q -x -p no:cacheprovider 2>&1 | tail -3
for rep in 1 2; do
  HIST_PIPE=0 timeout 600 python tools/time_history.py 400 400 16 1 f16 noref cx3 vm 2>>$OUT/err1.log | tee -a $OUT/hist.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pipe', d['pipelined_step'], 'chunks', d['chunks'], 'fused_ms', d['fused_ms'], 'warp_ms', d['warp_ms'])"
  for ch in 1 2 4 6 10 16 25 50; do
    HIST_CHUNKS=$ch timeout 600 python tools/time_history.py 400 400 16 1 f16 noref cx3 vm 2>>$OUT/err1.log | tee -a $OUT/hist.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pipe', d['pipelined_step'], 'chunks', d['chunks'], 'fused_ms', d['fused_ms'], 'warp_ms', d['warp_ms'])"
  done
done
timeout 600 python tools/time_history.py 400 400 16 1 bf16 noref cx3 vm 2>>$OUT/err1.log | tee -a $OUT/hist.jsonl | cut -c1-300
timeout 600 python tools/time_history.py 100 100 8 4 f16 cx3 vm 2>>$OUT/err1.log | tee -a $OUT/hist.jsonl | cut -c1-400
