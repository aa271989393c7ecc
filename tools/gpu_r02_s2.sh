#!/bin/bash
# Round-2 GPU session 2: where does the training route diverge / where does the S5 step time go.
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
FBBEV_EXPERIMENTAL=1 timeout -k 5 200 python -m pytest tests/test_gpu_conv3d.py -m gpu -q -p no:cacheprovider -k "training_route" -s > $OUT/s2_train_route.log 2>&1
grep -E "training-route|passed|failed|AssertionError" $OUT/s2_train_route.log | cut -c1-3000 | tail -8
cd /tmp
timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s2_prof_train -- python $REPO/tools/time_full.py train 2 f32 mfma > $OUT/s2_prof_train.log 2>&1
echo "rocprof train rc=$?"; tail -2 $OUT/s2_prof_train.log | cut -c1-600
cd $REPO
f=$(find $OUT/s2_prof_train -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-220
find $OUT -name "*.csv" -size +20M -delete
