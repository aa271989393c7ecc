#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
for dt in f32 bf16; do
  FBBEV_TRAIN_PROFILE=$OUT/r03_train_step_kernels_$dt.json timeout 600 python bench.py --mode train --steps 3 --warmup 2 --conv-dtype $dt > $OUT/train_$dt.json 2> $OUT/train_$dt.err; echo "train $dt rc=$?"; cut -c1-300 $OUT/train_$dt.json
done
timeout 300 python tools/time_fb.py BL2 4 30 4 > $OUT/fb_softmax_fused.json 2>$OUT/fb_softmax_fused.err; cut -c1-300 $OUT/fb_softmax_fused.json
timeout 900 python -m pytest tests/test_gpu_backward_projection.py -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_bp.log 2>&1; echo "pytest bp rc=$?"; tail -3 $OUT/pytest_bp.log | cut -c1-300
