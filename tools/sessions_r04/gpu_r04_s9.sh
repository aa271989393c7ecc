#!/bin/bash
# round 4, session 9: (key, value) pairs in the sort's intermediate array, on / off
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/r04_time_rank_pairs.jsonl
for pr in 1 0 1 0; do
  for cfg in "BL2 16" "REF 16" "BL2 4" "REF 4"; do
    FBBEV_RANK_PAIRS=$pr timeout 120 python tools/time_rank.py $cfg 2>/dev/null | sed "s/^{/{\"pairs\": $pr, /" >> $OUT/r04_time_rank_pairs.jsonl
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r04_time_rank_pairs.jsonl'):
    d=json.loads(l); print('pairs', d['pairs'], d['config'], d['B'], d['rank_build_ms'], d['checksum'][:2])
PY
cd /tmp; rm -rf $OUT/r04_prof_rank; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_prof_rank -- python $REPO/tools/time_rank.py BL2 16 > $OUT/r04_prof_rank.log 2>&1; cd $REPO
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r04_prof_rank/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_sort_scatter_seg' in r['Kernel_Name']]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
p0, p1 = d[0::2], d[1::2]
print('pass0 us', round(sorted(p0)[len(p0)//2], 1), 'pass1 us', round(sorted(p1)[len(p1)//2], 1))
f = glob.glob('gpurun_out/r04_prof_rank/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
