#!/bin/bash
# cross-attention block tail inside the sampler's (8-head) workgroups: A/B on one box
for t in 1 0 1 0; do
  echo "da_tail=$t BL2: $(FBBEV_FUSE_ATTN_TAIL_DA=$t python tools/time_fb.py BL2 4 50 4 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_fb"],4), round(d["ms_fb_graph"],4))')"
done
FBBEV_FUSE_ATTN_TAIL_DA=1 timeout 300 python -m pytest tests/test_gpu_backward_projection.py -q -m gpu -k "module or full or BackwardProjection or inference" 2>&1 | tail -2
