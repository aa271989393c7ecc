#!/bin/bash
# round 5, session 5: fused DA sampler -- the first staged copy of a coarse level requested before the offsets projection (A/B
# FBBEV_DA_FUSED_PRE); the whole GPU suite as a mid-round checkpoint
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/s05_time_fb.jsonl
for rep in 1 2; do
for knobs in "" "FBBEV_DA_FUSED_PRE=0" "FBBEV_DA_FUSED_STAGE=0"; do
  env $knobs timeout 300 python tools/time_fb.py BL2 4 40 4 2>/dev/null | sed "s/^{/{\"knobs\": \"$knobs\", /" >> $OUT/s05_time_fb.jsonl
done
done
python - <<'PY'
import json
for l in open('gpurun_out/s05_time_fb.jsonl'):
    d = json.loads(l); print(d['knobs'] or 'default', d['config'], d['B'], d['levels'], 'eager', round(d['ms_fb'], 4), 'graph', round(d['ms_fb_graph'], 4))
PY
cd /tmp
for tag in pre nopre; do
  rm -rf $OUT/s05_prof_$tag
  if [ $tag = nopre ]; then export FBBEV_DA_FUSED_PRE=0; else unset FBBEV_DA_FUSED_PRE; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s05_prof_$tag -- python $REPO/tools/time_fb.py BL2 4 30 4 > $OUT/s05_prof_$tag.log 2>&1; echo "rocprof $tag rc=$?"
done
unset FBBEV_DA_FUSED_PRE
cd $REPO
python - <<'PY'
import csv, glob
for tag in ('pre', 'nopre'):
    f = glob.glob(f'gpurun_out/s05_prof_{tag}/**/*kernel_stats.csv', recursive=True)[0]
    for r in list(csv.DictReader(open(f)))[:3]:
        print(tag, r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/s05_pytest_gpu.log 2>&1; echo "pytest(all gpu) rc=$?"
tail -4 $OUT/s05_pytest_gpu.log | cut -c1-300
grep -E "^E  |FAILED" $OUT/s05_pytest_gpu.log | head
find $OUT -name "*kernel_trace.csv" -size +20M -delete
