#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp
cd /tmp
for pr in 0 1; do
  rm -rf $OUT/prof_probe$pr; FBBEV_DA_PROBE=$pr timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_probe$pr -- python $REPO/tools/time_fb.py BL2 4 10 4 > /dev/null 2>&1
done
cd $REPO
python - <<'PY'
import csv, glob
for pr in (0, 1):
    f = glob.glob(f'gpurun_out/prof_probe{pr}/**/*kernel_stats.csv', recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if 'da_cross_attn_fused' in r['Name']: print('probe', pr, r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
