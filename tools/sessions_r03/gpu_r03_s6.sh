#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
T=0x2000000
rm -f $OUT/r03_pool_tolerance_mode.jsonl
# default tiling vs tolerance mode (flag word | SPLIT_LONG) at the shipped grid, BASELINE configs[0] (BL1) and configs[1] (BL2)
python tools/time_pool_flags.py REF 16 f32 64:$(printf "0x%x" $((0x24414 | T))) 128:0x24424 128:$(printf "0x%x" $((0x24424 | T))) >> $OUT/r03_pool_tolerance_mode.jsonl 2>&1
python tools/time_pool_flags.py BL1 4 f32 64:$(printf "0x%x" $((0x24414 | T))) 128:0x24424 128:$(printf "0x%x" $((0x24424 | T))) >> $OUT/r03_pool_tolerance_mode.jsonl 2>&1
python tools/time_pool_flags.py BL2 16 f32 128:$(printf "0x%x" $((0x24424 | T))) >> $OUT/r03_pool_tolerance_mode.jsonl 2>&1
grep -v amdgpu.ids $OUT/r03_pool_tolerance_mode.jsonl
FBBEV_TRAIN_PROFILE=$OUT/r03_train_step_kernels_bf16_clbn.json timeout 600 python bench.py --mode train --steps 3 --warmup 2 > $OUT/train_clbn.json 2> $OUT/train_clbn.err; echo "train rc=$?"; cut -c1-300 $OUT/train_clbn.json; tail -3 $OUT/train_clbn.err
timeout 1200 python -m pytest tests/test_gpu_conv3d.py tests/test_gpu_full_model.py tests/test_gpu_history.py -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_s6.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_s6.log | cut -c1-300
