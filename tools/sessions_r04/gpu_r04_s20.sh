#!/bin/bash
# round 4, session 20: kernel breakdown of S3 at the shipped shape (REF, B = 1 and B = 4)
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for B in 1 4; do
  rm -rf $OUT/r04_prof_ref_b$B; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_prof_ref_b$B -- python $REPO/tools/time_fb.py REF $B 30 1 > $OUT/r04_prof_ref_b$B.log 2>&1
done
cd $REPO
python - <<'PY'
import csv, glob
for B in (1, 4):
    f = glob.glob(f'gpurun_out/r04_prof_ref_b{B}/**/*kernel_stats.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    print('--- REF B =', B)
    for r in rows[:22]:
        print(r['Name'][:64], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
