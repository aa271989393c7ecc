#!/bin/bash
# training-step profile (BASELINE configs[3], 1 GPU) f32 and bf16 2-D stacks + the GPU tests that failed in session 2
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_backward_projection.py tests/test_gpu_parity.py -m gpu -q --timeout 900 -p no:cacheprovider -k "bound_of_fp64 or mixed_routes" > $OUT/pytest_fix.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_fix.log | cut -c1-300
cd /tmp
for dt in f32 bf16; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_$dt -- python $REPO/bench.py --mode train --steps 3 --warmup 2 --conv-dtype $dt > $OUT/train_$dt.json 2> $OUT/train_$dt.err; echo "train $dt rc=$?"; cut -c1-500 $OUT/train_$dt.json
done
cd $REPO
find $OUT -name "*kernel_trace.csv" -size +3M -delete
