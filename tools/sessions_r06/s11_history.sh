#!/bin/bash
# round 6, session 11: history fusion (S6) after the split-operand convolution kernel was rebuilt
REPO=$(pwd); OUT=$REPO/gpurun_out/s11; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_history.py -q -x -p no:cacheprovider 2>&1 | tail -3
for i in 1 2; do timeout 600 python tools/time_history.py 400 400 16 1 f16 noref cx3 vm 2>>$OUT/err1.log | tee -a $OUT/hist_cx3_vm.jsonl; done
timeout 600 python tools/time_history.py 400 400 16 1 bf16 noref cx3 vm 2>>$OUT/err1.log | tee -a $OUT/hist_cx3_vm.jsonl
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $REPO/tools/time_history.py 400 400 16 1 f16 noref cx3 vm > $OUT/prof.log 2>&1
cd $REPO
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); head -8 "$f" | cut -c1-220
timeout 600 python tools/scope_table.py $OUT/s6.json only_s6 2>&1 | tail -3 | cut -c1-400
