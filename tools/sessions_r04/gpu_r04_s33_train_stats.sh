#!/bin/bash
# kernel breakdown of the path's training step at the configs[2] pyramid and the shipped shape
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp
cd /tmp
for cfg in "BL2 4 4" "REF 4 1"; do
  tag=$(echo $cfg | tr ' ' '_')
  rm -rf $OUT/train_$tag; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_$tag -- python $REPO/tools/time_train.py $cfg > $OUT/train_$tag.log 2>&1
  cp $(ls $OUT/train_$tag/*/*kernel_stats.csv | head -1) $OUT/train_stats_$tag.csv
done
find $OUT -name "*kernel_trace.csv" -size +5M -delete
