#!/bin/bash
# round 4, session 7: XCD-contiguous chunk order of the sort's scatter passes: whole rank build, on / off, all shape classes
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/r04_time_rank_tuned.jsonl
for sw in 1; do
  for cfg in "BL2 16" "REF 16" "BL2 4" "REF 1" "BL1 1" "BL5 1" "REF 4"; do
    FBBEV_RANK_XCD_SWIZZLE=$sw timeout 120 python tools/time_rank.py $cfg 2>/dev/null | sed "s/^{/{\"xcd_swizzle\": $sw, /" >> $OUT/r04_time_rank_tuned.jsonl
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r04_time_rank_tuned.jsonl'):
    d=json.loads(l); print('swz', d['xcd_swizzle'], d['config'], d['B'], d['rank_build_ms'], d['checksum'][:2])
PY
cd /tmp; rm -rf $OUT/r04_prof_rank; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_prof_rank -- python $REPO/tools/time_rank.py BL2 16 > $OUT/r04_prof_rank.log 2>&1; cd $REPO
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r04_prof_rank/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
