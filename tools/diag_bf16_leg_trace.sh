#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp
cd /tmp; rm -rf $OUT/diag_trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/diag_trace -- python $REPO/tools/diag_bf16_leg.py > $OUT/diag_trace.log 2>&1
cd $REPO
tail -9 $OUT/diag_trace.log
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/diag_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
# split the trace into the three legs by big gaps in pool-kernel dtype: use thirds of time
T = int(rows[-1]['End_Timestamp']) - t0
agg = collections.defaultdict(lambda: [[], [], []])
pools = [r for r in rows if 'k_pool_fwd_dense2' in r['Kernel_Name']]
# leg boundaries: first bf16 pool kernel (template arg OT=1) and first f32 after it
b1 = next(int(r['Start_Timestamp']) for r in pools if ', 256, 1, true' in r['Kernel_Name'])
b2 = next(int(r['Start_Timestamp']) for r in pools if int(r['Start_Timestamp']) > b1 and ', 256, 0, false' in r['Kernel_Name'])
prev_end = None
gaps = [[], [], []]
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    leg = 0 if s < b1 else (1 if s < b2 else 2)
    agg[r['Kernel_Name'].split('(')[0][:60]][leg].append((e - s) / 1e3)
    if prev_end is not None: gaps[leg].append((s - prev_end) / 1e3)
    prev_end = e
for k, v in agg.items():
    if sum(len(x) for x in v) > 30:
        print(k, [(len(x), round(sum(x) / max(1, len(x)), 1), round(max(x) if x else 0, 1)) for x in v])
print('gaps mean/max us per leg', [(round(sum(g) / max(1, len(g)), 2), round(max(g), 1)) for g in gaps])
PY
find $OUT/diag_trace -name "*.csv" -size +20M -delete
