#!/bin/bash
# round 4, session 8: where does k_sort_scatter_seg spend its 35 us?  probes: 3 = prologue only, 2 = + pair loads, 1 = + ranking (no stores)
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for pr in 0 1 2 3; do
  rm -rf $OUT/r04_prof_rank_p$pr
  FBBEV_RANK_PROBE=$pr timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_prof_rank_p$pr -- python $REPO/tools/time_rank.py BL2 16 > $OUT/r04_prof_rank_p$pr.log 2>&1
done
cd $REPO
python - <<'PY'
import csv, glob
for pr in range(4):
    f = glob.glob(f'gpurun_out/r04_prof_rank_p{pr}/**/*kernel_trace.csv', recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if 'k_sort_scatter_seg' in r['Kernel_Name']]
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
    p0, p1 = d[0::2], d[1::2]
    print('probe', pr, 'pass0 us', round(sorted(p0)[len(p0)//2], 1), 'pass1 us', round(sorted(p1)[len(p1)//2], 1))
PY
