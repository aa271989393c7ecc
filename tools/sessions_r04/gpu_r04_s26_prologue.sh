#!/bin/bash
# fused DA kernel after the phase-A / fragment-prefetch rework: parity tests, probe (sample loops off) vs full, S3 timing
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_backward_projection.py -x -q -m gpu 2>&1 | tail -3
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
        if 'fused' in r['Name'] and 'history' not in r['Name']: print('probe', pr, r['Name'][:40], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
python tools/time_fb.py BL2 4 50 4 2>&1 | tail -4
