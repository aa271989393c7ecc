#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/time_train.py BL2 4 4 > $OUT/r03_time_train_BL2_B4_L4.json 2>$OUT/time_train.err; cat $OUT/r03_time_train_BL2_B4_L4.json | cut -c1-600
timeout 300 python tools/time_train.py REF 4 1 > $OUT/r03_time_train_REF_B4.json 2>>$OUT/time_train.err; cat $OUT/r03_time_train_REF_B4.json | cut -c1-400
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_path -- python $REPO/tools/time_train.py BL2 4 4 > $OUT/prof_train_path.log 2>&1
cd $REPO
python - <<'PY'
import csv,glob,shutil
f=sorted(glob.glob('gpurun_out/prof_train_path/**/*kernel_stats.csv',recursive=True))[-1]
shutil.copy(f,'gpurun_out/r03_rocprofv3_train_path_BL2_B4_L4.csv')
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:25]: print(r['Name'][:90].ljust(90), r['Calls'], r['AverageNs'], round(100*float(r['TotalDurationNs'])/tot,1))
PY
find $OUT -name "*kernel_trace.csv" -size +3M -delete
