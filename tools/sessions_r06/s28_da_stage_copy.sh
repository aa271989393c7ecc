#!/bin/bash
# round 6, session 28: the staged-plane copy of the one-kernel DA sampler requested in batches: kernel time by rocprofv3 (368-369 us before)
REPO=$(pwd); OUT=$REPO/gpurun_out/s28; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $REPO/tools/time_fb.py BL2 4 20 4 > $OUT/prof.log 2>&1
cd $REPO
python - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/s28/prof/**/*kernel_stats.csv',recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:6]: print(r['Name'][:50], r['Calls'], r['AverageNs'], r['MinNs'])
PY
for i in 1 2; do timeout 300 python tools/time_fb.py BL2 4 40 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('S3 eager', round(d['ms_fb'],4), 'graph', round(d.get('ms_fb_graph') or 0,4))"; done
timeout 1500 python -m pytest tests/test_gpu_backward_projection.py -q -x -p no:cacheprovider 2>&1 | tail -2
