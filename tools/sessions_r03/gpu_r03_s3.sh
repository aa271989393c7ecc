#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
rm -f $OUT/r03_exp_unit_sampler_patch.jsonl
for args in "160000 32 0 1 116 200" "160000 32 0 0 116 200" "40000 32 0 1 32 88" "10000 32 0 1 16 44" "160000 32 0 1 32 88" "160000 32 0 1 16 44"; do
  timeout 120 tools/micro/unit_sampler_pipeline $args >> $OUT/r03_exp_unit_sampler_patch.jsonl 2>&1
done
cat $OUT/r03_exp_unit_sampler_patch.jsonl
