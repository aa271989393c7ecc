#!/bin/bash
# round 6, session 9: same-box A/B of plain vs non-temporal stores for the backward projection's intermediates (FBBEV_STREAM_STORES build)
REPO=$(pwd); OUT=$REPO/gpurun_out/s09; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/time_fb.jsonl $OUT/train.jsonl
for rep in 1 2; do
for lib in fb_bev_amd/libfbbev_hip.so fb_bev_amd/_variants/libfbbev_hip_stream.so; do
  for cfg in "BL2 4 40 4" "REF 1 40 1" "REF 4 40 1"; do
    FBBEV_LIB=$REPO/$lib timeout 300 python tools/time_fb.py $cfg 2>/dev/null | sed "s/^{/{\"lib\": \"$(basename $lib)\", /" >> $OUT/time_fb.jsonl
  done
  FBBEV_LIB=$REPO/$lib timeout 300 python tools/train_path.py BL2 4 4 --steps 20 2>/dev/null | sed "s/^{/{\"lib\": \"$(basename $lib)\", /" >> $OUT/train.jsonl
done
done
python - <<'PY'
import json
for l in open('gpurun_out/s09/time_fb.jsonl'):
    d = json.loads(l); print(d['lib'], d['config'], d['B'], d['levels'], 'eager', round(d['ms_fb'], 4), 'graph', round(d.get('ms_fb_graph') or 0, 4))
for l in open('gpurun_out/s09/train.jsonl'):
    d = json.loads(l); print(d['lib'], 'train', round(d['ms_forward_backward_gradient_handed_over'], 3), 'fwd', round(d['ms_forward_train_mode'], 3))
PY
for lib in fb_bev_amd/libfbbev_hip.so fb_bev_amd/_variants/libfbbev_hip_stream.so; do
  rm -rf $OUT/prof
  cd /tmp && FBBEV_LIB=$REPO/$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $REPO/tools/time_fb.py BL2 4 20 4 > $OUT/prof.log 2>&1
  cd $REPO
  python - $lib <<'PY'
import csv, glob, sys
for f in glob.glob('gpurun_out/s09/prof/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(sys.argv[1][11:], '|', r['Name'][:60], r['Calls'], round(float(r['AverageNs']) / 1e3, 1))
PY
done
