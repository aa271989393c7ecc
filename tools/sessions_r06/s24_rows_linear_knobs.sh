#!/bin/bash
# round 6, session 24: k_rows_linear_x3 sensitivity to its row-tile knobs at the training path's sizes (R = 160 000 rows)
REPO=$(pwd); OUT=$REPO/gpurun_out/s24; mkdir -p $OUT; export TMPDIR=/tmp
for cfg in "default" "FBBEV_ROWS_LINEAR_RT=2" "FBBEV_ROWS_LINEAR_RT=4" "FBBEV_ROWS_LINEAR_RT=8" "FBBEV_ROWS_LINEAR_NT=1" "FBBEV_ROWS_LINEAR_NT=1 FBBEV_ROWS_LINEAR_RT=4"; do
  echo "== $cfg"
  if [ "$cfg" = "default" ]; then python tools/time_rows_kernels.py 2>/dev/null | grep rows_linear_x3 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['I'], '->', d['O'], round(d['us'], 1), 'us', round(d['TBps'], 2), 'TB/s; train_res', round(d['us_train_res'], 1))"
  else env $cfg python tools/time_rows_kernels.py 2>/dev/null | grep rows_linear_x3 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['I'], '->', d['O'], round(d['us'], 1), 'us', round(d['TBps'], 2), 'TB/s; train_res', round(d['us_train_res'], 1))"
  fi
done
