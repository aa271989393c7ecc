#!/bin/bash
# Round-2 GPU session 3: the one-launch-per-pass ranking chain + camera-keyed cache on hardware.
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 5 400 python -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider > $OUT/s3_parity.log 2>&1
echo "parity rc=$?"; tail -15 $OUT/s3_parity.log | cut -c1-400
timeout -k 5 200 python -m pytest tests/test_gpu_conv3d.py -m gpu -q -p no:cacheprovider -k "stacks" -s > $OUT/s3_stacks.log 2>&1
echo "stacks rc=$?"; grep -E "stack training routes|passed|failed|Error" $OUT/s3_stacks.log | cut -c1-1500 | tail -5
rm -f $OUT/s3_time_rank.jsonl
for c in "BL2 16" "BL2 1" "REF 16" "BL5 4"; do timeout -k 5 120 python tools/time_rank.py $c 2>>$OUT/s3_time_rank.err | tail -1 | tee -a $OUT/s3_time_rank.jsonl; done
timeout -k 5 300 python bench.py --steps 30 --warmup 5 > $OUT/s3_bench.json 2> $OUT/s3_bench.err; echo "bench rc=$?"; cat $OUT/s3_bench.json | cut -c1-1800; tail -3 $OUT/s3_bench.err
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s3_prof -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/s3_prof.log 2>&1
echo "rocprof rc=$?"
cd $REPO
f=$(find $OUT/s3_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-200
find $OUT -name "*.csv" -size +20M -delete
