#!/bin/bash
# round 6, session 10: read-ahead of the final pooling's gather sources (fbbev_touch), same-box A/B
REPO=$(pwd); OUT=$REPO/gpurun_out/s10; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/time_fb.jsonl $OUT/train.jsonl
for rep in 1 2; do
for ra in 0 1; do
  for cfg in "BL2 4 40 4" "REF 1 40 1" "REF 4 40 1" "BL2 1 40 4"; do
    FBBEV_POOL_READ_AHEAD=$ra timeout 300 python tools/time_fb.py $cfg 2>/dev/null | sed "s/^{/{\"read_ahead\": $ra, /" >> $OUT/time_fb.jsonl
  done
  FBBEV_POOL_READ_AHEAD=$ra timeout 300 python tools/train_path.py BL2 4 4 --steps 20 2>/dev/null | sed "s/^{/{\"read_ahead\": $ra, /" >> $OUT/train.jsonl
done
done
python - <<'PY'
import json
for l in open('gpurun_out/s10/time_fb.jsonl'):
    d = json.loads(l); print('read_ahead', d['read_ahead'], d['config'], d['B'], d['levels'], 'eager', round(d['ms_fb'], 4), 'graph', round(d.get('ms_fb_graph') or 0, 4))
for l in open('gpurun_out/s10/train.jsonl'):
    d = json.loads(l); print('read_ahead', d['read_ahead'], 'train', round(d['ms_forward_backward_gradient_handed_over'], 3), 'fwd', round(d['ms_forward_train_mode'], 3))
PY
timeout 600 python -m pytest tests/test_gpu_backward_projection.py -q -x -p no:cacheprovider -k "write_once or graphed" 2>&1 | tail -2
