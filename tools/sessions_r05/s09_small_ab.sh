#!/bin/bash
# round 5, session 9: cheap A/B knobs of S3 at configs[2]: hidden-chunk size of the tail + FFN kernel
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/s09_time_fb.jsonl
for rep in 1 2; do
for knobs in "" "FBBEV_TAIL_FFN_HC=64"; do
  for cfg in "BL2 4 40 4" "REF 4 40 1"; do
    env $knobs timeout 300 python tools/time_fb.py $cfg 2>/dev/null | sed "s/^{/{\"knobs\": \"$knobs\", /" >> $OUT/s09_time_fb.jsonl
  done
done
done
python - <<'PY'
import json
for l in open('gpurun_out/s09_time_fb.jsonl'):
    d = json.loads(l); print(d['knobs'] or 'default', d['config'], d['B'], d['levels'], 'eager', round(d['ms_fb'], 4), 'graph', round(d['ms_fb_graph'], 4))
PY
