#!/bin/bash
# Round-2 GPU session 7: 16-bit history ring + BASELINE configs[3]/[4] parity tests, fast-division interval kernel, bf16 train bench.
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 5 600 python -m pytest tests/test_gpu_history.py tests/test_gpu_full_model.py -m gpu -x -q -s -p no:cacheprovider > $OUT/s7_tests.log 2>&1
echo "tests rc=$?"; grep -E "configs\[|max rel err|passed|failed|Error|assert" $OUT/s7_tests.log | cut -c1-600 | tail -14
timeout -k 5 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider > $OUT/s7_parity.log 2>&1; echo "parity rc=$?"; tail -2 $OUT/s7_parity.log | cut -c1-200
rm -f $OUT/s7_time_rank.jsonl
for c in "BL2 16" "REF 16"; do timeout -k 5 120 python tools/time_rank.py $c 2>>$OUT/s7_time_rank.err | tail -1 | tee -a $OUT/s7_time_rank.jsonl; done
timeout -k 5 400 python bench.py --mode train --conv-dtype bf16 --steps 3 --warmup 2 > $OUT/s7_bench_train_bf16.json 2> $OUT/s7_bench_train_bf16.err; echo "bench train bf16 rc=$?"; cut -c1-400 $OUT/s7_bench_train_bf16.json; grep -v "MIOpen\|amdgpu.ids" $OUT/s7_bench_train_bf16.err | tail -4
