#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/s06; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train_path.py -q -x -s -p no:cacheprovider > $OUT/pytest_train.log 2>&1; echo "pytest rc=$?"; grep -E "Error|worst|forward|passed|failed|bits" $OUT/pytest_train.log | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_backward_projection.py -q -x -p no:cacheprovider -k "training or trainable or autocast or host_sync or write_once or fp64 or owned" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log | cut -c1-400
python tools/train_path.py BL2 4 4 --steps 20 --checksum --sites > $OUT/train.json 2> $OUT/err.log; echo "rc=$?"; tail -2 $OUT/err.log; python -c "
import json; d=json.load(open('$OUT/train.json')); print({k:v for k,v in d.items() if k not in ('grad_abs_sums','op_sites')})
tot=0
for r in d['op_sites']:
    n=r['op']
    if n.startswith('aten::') or n.endswith('Backward') or n in('EncoderLayerFn','ZMean','PoolAdd','TokenRows','BevQueries','RowsToNCHW') or n.startswith('autograd'): continue
    tot+=r['self_ms']
    if r['self_ms']>0.04: print('  %.3f %d %s'%(r['self_ms'], r['calls'], r['op'][:100]))
print(tot)
"
