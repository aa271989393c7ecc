#!/bin/bash
# DA backward: unit gradients on head planes against the row kernel -- parity tests, kernel times, training step
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -k "bwd or backward or grad or train or owned" -p no:cacheprovider 2>&1 | tail -3
for up in 1 0; do
  echo "unit_planes=$up: $(FBBEV_DA_BWD_UNIT_PLANES=$up python tools/time_train.py BL2 4 4 2>/dev/null | tail -1 | cut -c1-330)"
done
cd /tmp
for up in 1 0; do
  rm -rf $OUT/prof_up; FBBEV_DA_BWD_UNIT_PLANES=$up timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_up -- python $REPO/tools/time_train.py BL2 4 4 > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob('$OUT/prof_up/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('bwd_unit', 'unit_planes', 'rows_to_head', 'scatter_owned', 'hitlist')): print('up $up', r['Name'][:50], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
done
