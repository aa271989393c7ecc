#!/bin/bash
# round 4, session 23: how many z groups pay where (partial buffer traffic vs chain length)
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/r04_zmean_zgroups.jsonl
for zg in 1 2 4 8; do
  for cfg in "REF 4 1" "BL2 1 4" "REF 16 1"; do
    set -- $cfg
    FBBEV_ZMEAN_ZGROUPS=$zg timeout 300 python tools/time_fb.py $1 $2 40 $3 2>/dev/null | sed "s/^{/{\"zmean_zgroups\": $zg, /" >> $OUT/r04_zmean_zgroups.jsonl
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r04_zmean_zgroups.jsonl'):
    d=json.loads(l); print('zg', d['zmean_zgroups'], d['config'], d['B'], 'fb', round(d['ms_fb'],4), 'graph', round(d['ms_fb_graph'],4))
PY
