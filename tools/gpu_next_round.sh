#!/bin/bash
# First GPU session of the next round: the items this round's GPU budget did not reach.
#   gpurun --timeout 900 -- 'bash tools/gpu_next_round.sh'
# 1. experimental parity tests of the fp32-MFMA conv3d / blend kernels (only emulator-validated so far)
# 2. S4 (full forward) timing: vendor fp32, vendor bf16, MFMA route      3. S5 (full training step) timing
# Every step has its own timeout and log under gpurun_out/; nothing here is part of the default GPU suite.
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
FBBEV_EXPERIMENTAL=1 PYTHONFAULTHANDLER=1 timeout -k 5 200 python -m pytest tests/test_gpu_conv3d.py tests/test_gpu_bevdet.py -m gpu -q -p no:cacheprovider > $OUT/conv3d_tests.log 2>&1
echo "conv3d tests rc=$?"; grep -E "passed|failed|Error|crashed" $OUT/conv3d_tests.log | tail -5
for mode in "1 f32" "1 bf16" "1 bf16 tune" "1 f32 mfma" "1 f32 mfma_bf16" "1 f32 mfma_bf16_tiled" "4 bf16" "4 f32 mfma" "4 f32 mfma_bf16"; do
  timeout -k 5 120 python tools/time_full.py infer $mode 2>> $OUT/time_full.err | tail -1 | tee -a $OUT/time_full.jsonl
done
for mode in "2 f32" "2 f32 mfma" "4 bf16"; do
  timeout -k 5 240 python tools/time_full.py train $mode 2>> $OUT/time_full.err | tail -1 | tee -a $OUT/time_full.jsonl
done
cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_full -- python $REPO/tools/time_full.py infer 1 f32 mfma > $OUT/prof_full.log 2>&1
echo "rocprof mfma rc=$?"
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_full_vendor -- python $REPO/tools/time_full.py infer 1 bf16 > $OUT/prof_full_vendor.log 2>&1
echo "rocprof vendor rc=$?"
find $OUT -name "*.csv" -size +20M -delete
