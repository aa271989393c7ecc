#!/bin/bash
# round 6, session 25: k_rows_linear_x3p (persistent row-linear kernel) against k_rows_linear_x3: per-shape times, S3, the training step
REPO=$(pwd); OUT=$REPO/gpurun_out/s25; mkdir -p $OUT; export TMPDIR=/tmp
PY='
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d["I"], "->", d["O"], round(d["us"], 1), "us", round(d["TBps"], 2), "TB/s; train_res", round(d["us_train_res"], 1), "train_mask", round(d["us_train_mask"], 1))'
for p in 0 1; do echo "== FBBEV_ROWS_LINEAR_P=$p"; FBBEV_ROWS_LINEAR_P=$p python tools/time_rows_kernels.py 2>/dev/null | grep '"rows_linear_x3"' | python -c "$PY"; done
for rep in 1 2; do for p in 0 1; do
  FBBEV_ROWS_LINEAR_P=$p timeout 300 python tools/time_fb.py BL2 4 40 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('P=$p S3 eager', round(d['ms_fb'],4), 'graph', round(d.get('ms_fb_graph') or 0,4))"
  FBBEV_ROWS_LINEAR_P=$p timeout 300 python tools/train_path.py BL2 4 4 --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('P=$p train', round(d['ms_forward_backward_gradient_handed_over'],3))"
done; done
timeout 1500 python -m pytest tests/test_gpu_backward_projection.py tests/test_gpu_train_path.py tests/test_gpu_block_kernels.py -q -x -p no:cacheprovider 2>&1 | tail -2
