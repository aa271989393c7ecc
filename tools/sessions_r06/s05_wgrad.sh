#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/s05; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train_path.py -q -x -s -p no:cacheprovider > $OUT/pytest_train.log 2>&1; echo "pytest rc=$?"; grep -E "Error|worst|forward|passed|failed|bits" $OUT/pytest_train.log | cut -c1-400
for w in 1 0; do
FBBEV_TRAIN_WGRAD=$w python tools/train_path.py BL2 4 4 --steps 20 --checksum --sites > $OUT/train_w$w.json 2> $OUT/err_w$w.log; echo "rc=$?"; tail -2 $OUT/err_w$w.log; python -c "
import json; d=json.load(open('$OUT/train_w$w.json')); print({k:v for k,v in d.items() if k not in ('grad_abs_sums','op_sites')})
for r in d['op_sites']:
    if 'wgrad' in r['op'] or 'Cijk' in r['op'] or 'reduce_kernel' in r['op']: print('  %.3f %d %s'%(r['self_ms'], r['calls'], r['op'][:90]))
"
done
