#!/bin/bash
# round 6, session 29: S6 (scope table protocol) against the warp's rows per band, same box
REPO=$(pwd); OUT=$REPO/gpurun_out/s29; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do for yb in 8 16 32; do
  FBBEV_HISTORY_VM_YB=$yb timeout 600 python tools/scope_table.py $OUT/s6_$yb.json only_s6 2>&1 | grep "x3" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('YB=$yb', d['ms_p10_p50_p90'])"
done; done
