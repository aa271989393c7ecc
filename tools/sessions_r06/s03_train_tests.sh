#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/s03; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train_path.py -q -x -s -p no:cacheprovider > $OUT/pytest_train.log 2>&1; echo "pytest rc=$?"; grep -v "^$" $OUT/pytest_train.log | tail -25 | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_backward_projection.py -q -x -s -p no:cacheprovider -k "training or trainable or autocast or host_sync or write_once or fp64 or owned" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log | cut -c1-400
python tools/train_path.py BL2 4 4 --steps 20 --checksum > $OUT/train.json 2> $OUT/err.log; echo "rc=$?"; tail -3 $OUT/err.log; python -c "
import json; d=json.load(open('$OUT/train.json')); print({k:v for k,v in d.items() if k not in ('grad_abs_sums',)})"
