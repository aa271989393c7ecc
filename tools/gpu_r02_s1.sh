#!/bin/bash
# Round-2 GPU session 1: run the tests that were env-gated in round 1, then S4 / S5 timings of the
# detector with the vendor convolution route and with the hand-written route.  Every step has its own timeout.
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
FBBEV_EXPERIMENTAL=1 PYTHONFAULTHANDLER=1 timeout -k 5 300 python -m pytest tests/test_gpu_conv3d.py tests/test_gpu_bevdet.py -m gpu -q -p no:cacheprovider > $OUT/s1_gated_tests.log 2>&1
echo "gated tests rc=$?"; tail -30 $OUT/s1_gated_tests.log
rm -f $OUT/s1_time_full.jsonl
for mode in "infer 1 bf16" "infer 1 f32 mfma" "infer 1 f32 mfma_bf16" "infer 1 f32 mfma_bf16_tiled" "train 2 f32" "train 2 f32 mfma" "train 4 bf16"; do
  timeout -k 5 150 python tools/time_full.py $mode 2>> $OUT/s1_time_full.err | tail -1 | tee -a $OUT/s1_time_full.jsonl
done
tail -5 $OUT/s1_time_full.err
