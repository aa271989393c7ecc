#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
FBBEV_TRAIN_PROFILE_COPIES=1 FBBEV_TRAIN_PROFILE=$OUT/r03_train_step_kernels_bf16_v3.json timeout 900 python bench.py --mode train --steps 3 --warmup 2 > $OUT/train_v3.json 2> $OUT/train_v3.err; echo "train rc=$?"; cut -c1-200 $OUT/train_v3.json; tail -2 $OUT/train_v3.err
rm -f $OUT/r03_time_rank_shapes.jsonl
for sh in "" "6,6,2" "6,5,2" "5,5,1" "4,6,2" "6,2,2"; do
  FBBEV_RANK_SHAPE=$sh timeout 120 python tools/time_rank.py BL2 16 2>/dev/null | sed "s/^{/{\"shape\": \"$sh\", /" >> $OUT/r03_time_rank_shapes.jsonl
done
for sh in "" "6,6,2" "5,5,1" "6,5,1"; do
  FBBEV_RANK_SHAPE=$sh timeout 120 python tools/time_rank.py REF 16 2>/dev/null | sed "s/^{/{\"shape\": \"$sh\", /" >> $OUT/r03_time_rank_shapes.jsonl
done
cat $OUT/r03_time_rank_shapes.jsonl | cut -c1-200
T=0x2000000
python tools/time_pool_flags.py REF 16 f32 64:$(printf "0x%x" $((0x24414 | T))) 2>/dev/null | cut -c1-200
python tools/time_pool_flags.py BL1 4 f32 64:$(printf "0x%x" $((0x24414 | T))) 2>/dev/null | cut -c1-200
python tools/time_pool_flags.py BL2 16 f32 128:$(printf "0x%x" $((0x24424 | T))) 2>/dev/null | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_gpu.log | cut -c1-300
