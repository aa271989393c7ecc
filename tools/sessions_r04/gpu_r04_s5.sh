#!/bin/bash
# round 4, session 5: samples in flight per lane of the one-kernel DA sampler (2 / 3 / 4)
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/r04_time_fb_np.jsonl
for np in 2 3 4 2 3 4; do
  FBBEV_DA_FUSED_NP=$np timeout 300 python tools/time_fb.py BL2 4 50 4 2>/dev/null | sed "s/^{/{\"np\": $np, /" >> $OUT/r04_time_fb_np.jsonl
done
for np in 2 3; do
  FBBEV_DA_FUSED_NP=$np timeout 300 python tools/time_fb.py REF 4 50 1 2>/dev/null | sed "s/^{/{\"np\": $np, /" >> $OUT/r04_time_fb_np.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r04_time_fb_np.jsonl'):
    d=json.loads(l); print('np', d['np'], d['config'], d['B'], 'fb', round(d['ms_fb'],4), 'graph', round(d['ms_fb_graph'],4))
PY
