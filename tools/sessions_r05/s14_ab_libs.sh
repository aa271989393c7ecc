#!/bin/bash
# round 5, session 14: same-box A/B of two builds of the library (FBBEV_LIB): S3 timing + kernel trace per build
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/s14_time_fb.jsonl
for rep in 1 2; do
for lib in fb_bev_amd/_ab_head.so fb_bev_amd/libfbbev_hip.so; do
  for cfg in "BL2 4 40 4" "REF 4 40 1"; do
    FBBEV_LIB=$REPO/$lib timeout 300 python tools/time_fb.py $cfg 2>/dev/null | sed "s/^{/{\"knobs\": \"$(basename $lib)\", /" >> $OUT/s14_time_fb.jsonl
  done
done
done
python - <<'PY'
import json
for l in open('gpurun_out/s14_time_fb.jsonl'):
    d = json.loads(l); print(d['knobs'] or 'default', d['config'], d['B'], d['levels'], 'eager', round(d['ms_fb'], 4), 'graph', round(d['ms_fb_graph'], 4))
PY
for lib in fb_bev_amd/_ab_head.so fb_bev_amd/libfbbev_hip.so; do
  rm -rf $OUT/s14_prof
  cd /tmp && FBBEV_LIB=$REPO/$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s14_prof -- python $REPO/tools/time_fb.py BL2 4 20 4 > $OUT/s14_prof.log 2>&1
  cd $REPO
  python - $lib <<'PY'
import csv, glob, sys
for f in glob.glob('gpurun_out/s14_prof/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:9]:
        print(sys.argv[1][11:], '|', r['Name'][:70], r['Calls'], round(float(r['AverageNs']) / 1e3, 1))
PY
done
