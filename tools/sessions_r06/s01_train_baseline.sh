#!/bin/bash
# round 6, session 1: where the path's training step spends its time at HEAD (gradient handed over), before any change
REPO=$(pwd); OUT=$REPO/gpurun_out/s01; mkdir -p $OUT; export TMPDIR=/tmp
python tools/train_path.py BL2 4 4 --steps 20 --sites --checksum --no-feat-grad > $OUT/train_nofeatgrad.json 2> $OUT/err1.log; cut -c1-600 $OUT/train_nofeatgrad.json
python tools/train_path.py BL2 4 4 --steps 20 --sites --checksum > $OUT/train.json 2> $OUT/err2.log; cut -c1-600 $OUT/train.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $REPO/tools/train_path.py BL2 4 4 --profile-steps 10 > $OUT/prof.log 2>&1; echo "rocprof rc=$?"
cd $REPO
find $OUT -name "*kernel_stats.csv" | head; find $OUT -name "*.csv" -size +20M -delete
