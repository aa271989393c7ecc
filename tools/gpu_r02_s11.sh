#!/bin/bash
# Round-2 GPU session 11: LDS-staged history warp, raw-fragment 16-bit history conv.
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 5 400 python -m pytest tests/test_gpu_history.py -m gpu -x -q -p no:cacheprovider > $OUT/s11_tests.log 2>&1
echo "tests rc=$?"; tail -4 $OUT/s11_tests.log | cut -c1-300
rm -f $OUT/s11_hist.jsonl
for mode in lds direct; do
  for a in "100 100 8 1 f32" "100 100 8 1 f16 noref" "200 200 16 4 f32 noref" "400 400 16 1 f16 noref" "400 400 16 1 f32 noref"; do
    FBBEV_HISTORY_WARP=$mode timeout -k 5 200 python tools/time_history.py $a 2>>$OUT/s11_hist.err | tail -1 | sed "s/^{/{\"warp\": \"$mode\", /" | tee -a $OUT/s11_hist.jsonl | cut -c1-330
  done
done
