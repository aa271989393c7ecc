#!/bin/bash
# round 4, session 18: hidden chunk of the one-kernel FFN: 64 units (2 workgroups / CU) vs 32 (3 workgroups / CU)
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/r04_time_fb_ffn_hc.jsonl
for hc in 32 64 32 64; do
  FBBEV_FFN_HC=$hc timeout 300 python tools/time_fb.py BL2 4 50 4 2>/dev/null | sed "s/^{/{\"ffn_hc\": $hc, /" >> $OUT/r04_time_fb_ffn_hc.jsonl
  FBBEV_FFN_HC=$hc timeout 300 python tools/time_fb.py REF 4 50 1 2>/dev/null | sed "s/^{/{\"ffn_hc\": $hc, /" >> $OUT/r04_time_fb_ffn_hc.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r04_time_fb_ffn_hc.jsonl'):
    d=json.loads(l); print('ffn_hc', d['ffn_hc'], d['config'], d['B'], 'fb', round(d['ms_fb'],4), 'graph', round(d['ms_fb_graph'],4))
PY
