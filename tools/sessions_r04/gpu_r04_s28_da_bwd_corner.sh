#!/bin/bash
# DA / MSDA backward after the corner-outer interchange: parity tests, per-launch durations, training step
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -k "bwd or backward or grad or train" -p no:cacheprovider 2>&1 | tail -3
cd /tmp
rm -rf $OUT/dabwd_trace; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/dabwd_trace -- python $REPO/tools/time_train.py BL2 4 4 > $OUT/dabwd_trace.log 2>&1
cd $REPO
python - <<'PY'
import csv, glob, collections, json
f = sorted(glob.glob('gpurun_out/dabwd_trace/**/*kernel_trace.csv', recursive=True), key=lambda p: -__import__('os').path.getmtime(p))[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = ('k_da_cross_attn_bwd_scatter', 'k_da_bwd_reduce', 'k_da_bwd_hitinfo', 'k_da_cross_attn_bwd_unit', 'k_msda_bwd', 'k_da_bwd_')
per = collections.defaultdict(list); seq = 0
for r in rows:
    n = r['Kernel_Name']
    for k in names:
        if k in n:
            d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            per[n.split('(')[0][-60:]].append(d)
            break
tot = 0
for k, v in per.items():
    v = v[len(v) // 3:]
    print(k, len(v), round(sum(v) / len(v), 1))
PY
python tools/time_train.py BL2 4 4 2>/dev/null | tail -1 | cut -c1-300
python tools/time_train.py REF 4 1 2>/dev/null | tail -1 | cut -c1-300
