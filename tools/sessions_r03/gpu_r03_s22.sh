#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/r03_time_fb_REF_variants.jsonl
for v in "" "FBBEV_DA_PIPE=0" "FBBEV_DA_PATCH=0"; do
  for B in 1 4; do
    env $v timeout 300 python tools/time_fb.py REF $B 50 1 2>/dev/null | sed "s/^{/{\"env\": \"$v\", /" >> $OUT/r03_time_fb_REF_variants.jsonl
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r03_time_fb_REF_variants.jsonl'):
    d=json.loads(l); print(d['env'].ljust(18), d['B'], 'fb', round(d['ms_fb'],4), 'graph', d.get('ms_fb_graph'))
PY
timeout 300 python tools/time_train.py REF 4 1 sites > $OUT/r03_time_train_REF_B4_sites.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_time_train_REF_B4_sites.json'))
print(d['ms_forward_backward'])
for r in d['op_sites'][:14]: print(round(r['self_ms'],3), r['calls'], r['op'][:80])
PY
