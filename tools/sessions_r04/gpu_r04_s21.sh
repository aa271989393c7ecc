#!/bin/bash
# round 4, session 21: channel groups of k_pool_zmean (it walks the Z planes serially per workgroup)
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/r04_zmean_csplit.jsonl
for cs in 0 2 5 10; do
  for cfg in "REF 1 1" "REF 4 1" "BL2 4 4"; do
    set -- $cfg
    FBBEV_ZMEAN_CSPLIT=$cs timeout 300 python tools/time_fb.py $1 $2 50 $3 2>/dev/null | sed "s/^{/{\"zmean_csplit\": $cs, /" >> $OUT/r04_zmean_csplit.jsonl
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r04_zmean_csplit.jsonl'):
    d=json.loads(l); print('zmean_csplit', d['zmean_csplit'], d['config'], d['B'], 'fb', round(d['ms_fb'],4), 'graph', round(d['ms_fb_graph'],4))
PY
