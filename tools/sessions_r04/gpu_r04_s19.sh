#!/bin/bash
# round 4, session 19: camera-token pyramid in one launch: module tests + S3
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward_projection.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -2
for i in 1 2; do
  timeout 300 python tools/time_fb.py BL2 4 50 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config'], d['B'], 'fb', round(d['ms_fb'],4), 'graph', round(d['ms_fb_graph'],4))"
  timeout 300 python tools/time_fb.py REF 4 50 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config'], d['B'], 'fb', round(d['ms_fb'],4), 'graph', round(d['ms_fb_graph'],4))"
done
