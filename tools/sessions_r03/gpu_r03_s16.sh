#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/time_train.py BL2 4 4 sites > $OUT/r03_time_train_BL2_B4_L4_sites.json 2>$OUT/time_train.err; tail -3 $OUT/time_train.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_time_train_BL2_B4_L4_sites.json'))
print(d['ms_forward_backward'])
for r in d['op_sites'][:40]: print(round(r['self_ms'],3), r['calls'], r['op'], r['shapes'][:110], [s.split('/')[-1][:60] for s in r['stack'][:3]])
PY
