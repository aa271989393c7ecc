#!/bin/bash
# owned-plane scatter: 1024-thread workgroups, plane KB
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp
cd /tmp
run() {
  rm -rf $OUT/own_r; env "$@" timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/own_r -- python $REPO/tools/time_train.py BL2 4 4 > /dev/null 2>&1
  python - "$*" <<PY
import csv, glob, sys
f = glob.glob('$OUT/own_r/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'scatter_owned' in r['Name']: print(sys.argv[1], '|', r['Name'][:44], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
}
run FBBEV_DA_BWD_THREADS=512
run FBBEV_DA_BWD_THREADS=1024
run FBBEV_DA_BWD_THREADS=1024 FBBEV_DA_BWD_LDS_KB=144
run FBBEV_DA_BWD_THREADS=512 FBBEV_DA_BWD_LDS_KB=144
run FBBEV_DA_BWD_THREADS=1024 FBBEV_DA_BWD_LDS_KB=72
run FBBEV_DA_BWD_THREADS=512 FBBEV_DA_BWD_LDS_KB=72
