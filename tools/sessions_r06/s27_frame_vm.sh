#!/bin/bash
# round 6, session 27: k_history_frame_vm with its plane pieces requested ten at a time
REPO=$(pwd); OUT=$REPO/gpurun_out/s27; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $REPO/tools/time_history.py 400 400 16 1 f16 noref cx3 vm > $OUT/prof.log 2>&1
cd $REPO
python - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/s27/prof/**/*kernel_stats.csv',recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:10]:
    if 'history' in r['Name']: print(r['Name'][:44], r['Calls'], r['AverageNs'], r['MinNs'])
PY
timeout 900 python -m pytest tests/test_gpu_history.py -q -x -p no:cacheprovider 2>&1 | tail -2
timeout 600 python tools/scope_table.py $OUT/s6.json only_s6 2>&1 | grep "x3" | cut -c1-260
