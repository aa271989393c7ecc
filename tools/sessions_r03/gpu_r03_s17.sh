#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward_projection.py -m gpu -q --timeout 900 -p no:cacheprovider -x > $OUT/pytest_bp.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_bp.log | cut -c1-300
timeout 300 python tools/time_train.py BL2 4 4 sites > $OUT/r03_time_train_BL2_B4_L4_sites_splitk.json 2>$OUT/time_train.err; tail -2 $OUT/time_train.err
timeout 300 python tools/time_train.py REF 4 1 2>/dev/null | cut -c1-300
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_time_train_BL2_B4_L4_sites_splitk.json'))
print(d['ms_forward_backward'], d['ms_forward_train_mode'])
for r in d['op_sites'][:32]: print(round(r['self_ms'],3), r['calls'], r['op'][:70], r['shapes'][:100])
PY
