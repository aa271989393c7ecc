#!/bin/bash
# DA backward at the configs[2] pyramid: per-launch durations of the region launches + LDS counters per region
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/dabwd_trace; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/dabwd_trace -- python $REPO/tools/time_train.py BL2 4 4 > $OUT/dabwd_trace.log 2>&1
i=0
for set in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1)); rm -rf $OUT/dabwd_pmc$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/dabwd_pmc$i -- python $REPO/tools/time_train.py BL2 4 4 > $OUT/dabwd_pmc$i.log 2>&1; echo "pass $i rc=$?"
done
cd $REPO
python - <<'PY'
import csv, glob, collections, json
f = glob.glob('gpurun_out/dabwd_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = ('k_da_cross_attn_bwd_scatter', 'k_da_bwd_reduce', 'k_da_bwd_hitinfo', 'k_da_cross_attn_bwd_unit', 'k_msda_bwd')
per = collections.defaultdict(list); seq = 0
for r in rows:
    n = r['Kernel_Name']
    for k in names:
        if k in n:
            d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            if k == 'k_da_cross_attn_bwd_scatter':
                per[f'scatter[{seq % 6}] grid={r.get("Grid_Size_X", r.get("Grid_Size"))} lds={r.get("LDS_Block_Size", r.get("LDS_Block_Size_Bytes"))}'].append(d); seq += 1
            else:
                per[k].append(d)
out = {}
for k, v in per.items():
    v = v[len(v) // 3:]
    out[k] = round(sum(v) / len(v), 1); print(k, len(v), out[k])
pm = collections.defaultdict(lambda: collections.defaultdict(list))
for i in range(1, 5):
    fs = glob.glob(f'gpurun_out/dabwd_pmc{i}/**/*counter_collection.csv', recursive=True)
    if not fs: print('no counters in pass', i); continue
    rs = [r for r in csv.DictReader(open(fs[0])) if 'k_da_cross_attn_bwd_scatter' in r['Kernel_Name']]
    disp = sorted({int(r['Dispatch_Id']) for r in rs})
    idx = {d: j % 6 for j, d in enumerate(disp)}
    for r in rs:
        pm[idx[int(r['Dispatch_Id'])]][r['Counter_Name']].append(float(r['Counter_Value']))
res = {str(k): {c: round(sum(x) / len(x), 1) for c, x in v.items()} for k, v in sorted(pm.items())}
for k, v in res.items(): print('region', k, v)
json.dump({'durations_us': out, 'scatter_counters_per_region': res}, open('gpurun_out/dabwd_diag.json', 'w'), indent=1)
PY
find $OUT -name "*.csv" -size +5M -delete
