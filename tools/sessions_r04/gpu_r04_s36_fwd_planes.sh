#!/bin/bash
# training forward on head planes: parity tests, kernel times, training step
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -k "bwd or backward or grad or train or owned or module" -p no:cacheprovider 2>&1 | tail -3
for fp in 1 0; do
  echo "fwd_planes=$fp: $(FBBEV_DA_FWD_PLANES=$fp python tools/time_train.py BL2 4 4 2>/dev/null | tail -1 | cut -c1-330)"
done
echo "REF: $(python tools/time_train.py REF 4 1 2>/dev/null | tail -1 | cut -c1-330)"
cd /tmp
rm -rf $OUT/prof_up; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_up -- python $REPO/tools/time_train.py BL2 4 4 > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob('$OUT/prof_up/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('fwd_planes', 'fwd_unit', 'unit_planes', 'rows_to_head', 'scatter_owned', 'hitlist')): print(r['Name'][:50], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
