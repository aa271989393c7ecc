#!/bin/bash
# Round-2 GPU session 6: full GPU suite on the new chain, rank timing + PMC, train-mode bench smoke run.
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/s6_pytest_gpu.log 2>&1
echo "pytest gpu rc=$?"; tail -6 $OUT/s6_pytest_gpu.log | cut -c1-300
rm -f $OUT/s6_time_rank.jsonl
for c in "BL2 16" "BL2 1" "BL2 4" "REF 16" "BL5 4"; do timeout -k 5 120 python tools/time_rank.py $c 2>>$OUT/s6_time_rank.err | tail -1 | tee -a $OUT/s6_time_rank.jsonl; done
timeout -k 5 300 python bench.py --steps 30 --warmup 5 > $OUT/s6_bench.json 2> $OUT/s6_bench.err; echo "bench rc=$?"; cut -c1-700 $OUT/s6_bench.json
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s6_prof -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/s6_prof.log 2>&1
echo "rocprof rc=$?"
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/s6_pmc -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/s6_pmc.log 2>&1
echo "pmc rc=$?"
cd $REPO
python - <<'P'
import csv,glob,collections
f=glob.glob('gpurun_out/s6_prof/*/*_kernel_stats.csv')
if f:
    for r in csv.DictReader(open(f[0])):
        print(r['Name'][:44], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
f=glob.glob('gpurun_out/s6_pmc/*/*counter_collection.csv')
if f:
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        agg[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
P
timeout -k 5 400 python bench.py --mode train --steps 3 --warmup 2 > $OUT/s6_bench_train.json 2> $OUT/s6_bench_train.err; echo "bench train rc=$?"; cut -c1-1500 $OUT/s6_bench_train.json; grep -v "MIOpen\|amdgpu.ids" $OUT/s6_bench_train.err | tail -5
find $OUT -name "*.csv" -size +20M -delete
