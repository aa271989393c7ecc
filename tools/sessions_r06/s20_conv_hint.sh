#!/bin/bash
# round 6, session 20: split-operand convolutions with the fragment reads requested a set ahead (sched_group_barrier hints): kernel time
REPO=$(pwd); OUT=$REPO/gpurun_out/s20; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp && HIST_FUSED_X3=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $REPO/tools/time_history.py 400 400 16 1 f16 noref cx3 vm > $OUT/prof.log 2>&1
cd $REPO
python - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/s20/prof/**/*kernel_stats.csv',recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:8]:
    if 'history' in r['Name']: print(r['Name'][:44], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
timeout 900 python -m pytest tests/test_gpu_history.py -q -x -p no:cacheprovider 2>&1 | tail -4
