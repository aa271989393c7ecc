#!/bin/bash
# round 6, session 26: rows per band of the voxel-major warp (FBBEV_HISTORY_VM_YB) at 32 taps in flight per thread
REPO=$(pwd); OUT=$REPO/gpurun_out/s26; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do for yb in 2 4 8 16 32; do
  FBBEV_HISTORY_VM_YB=$yb timeout 600 python tools/time_history.py 400 400 16 1 f16 noref cx3 vm 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('YB=$yb step', d['fused_ms'], 'warp', d['warp_ms'])"
done; done
