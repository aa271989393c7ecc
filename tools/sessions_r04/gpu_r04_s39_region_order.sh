REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp
cd /tmp
for bf in 0 1 0 1; do
  rm -rf $OUT/own_r; FBBEV_DA_BWD_BIG_FIRST=$bf timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/own_r -- python $REPO/tools/time_train.py BL2 4 4 > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob('$OUT/own_r/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'scatter_owned' in r['Name']: print('big_first $bf', r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
done
