#!/bin/bash
# owned-plane scatter: one region at a time (temporary probe), 512 / 256 threads
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp
cd /tmp
for r in 0 1 2 3 4 5 6 all; do
  rm -rf $OUT/own_r; if [ $r != all ]; then export FBBEV_DA_BWD_ONLY_REGION=$r; else unset FBBEV_DA_BWD_ONLY_REGION; fi
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/own_r -- python $REPO/tools/time_train.py BL2 4 4 > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob('$OUT/own_r/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'scatter_owned' in r['Name']: print('region $r', r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
done
unset FBBEV_DA_BWD_ONLY_REGION
for t in 256; do
  rm -rf $OUT/own_r; FBBEV_DA_BWD_THREADS=$t timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/own_r -- python $REPO/tools/time_train.py BL2 4 4 > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob('$OUT/own_r/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'scatter_owned' in r['Name']: print('threads $t', r['Name'][:40], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
done
