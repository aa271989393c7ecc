#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/s08; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -s -p no:cacheprovider -k "lidar_coor" 2>&1 | grep -E "get_lidar|passed|failed|Error" | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 5 --no-fb-projection --no-cpu-baseline --no-alt-storage > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -2 $OUT/bench.err
python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step']); print(d.get('reference_same_gpu'))"
