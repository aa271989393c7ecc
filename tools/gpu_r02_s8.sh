#!/bin/bash
# Round-2 GPU session 8: configs[3]/[4] tests, training step after channel padding (bench + profile), scope table, PMC passes.
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 5 600 python -m pytest tests/test_gpu_history.py tests/test_gpu_full_model.py tests/test_gpu_conv3d.py -m gpu -q -s -p no:cacheprovider > $OUT/s8_tests.log 2>&1
echo "tests rc=$?"; grep -E "configs\[|max rel err|passed|failed|Error|^E  " $OUT/s8_tests.log | cut -c1-700 | tail -14
timeout -k 5 400 python bench.py --mode train --steps 4 --warmup 2 > $OUT/s8_bench_train.json 2> $OUT/s8_bench_train.err; echo "bench train rc=$?"; cut -c1-330 $OUT/s8_bench_train.json
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s8_prof_train -- python $REPO/bench.py --mode train --steps 3 --warmup 2 > $OUT/s8_prof_train.log 2>&1
echo "rocprof train rc=$?"
cd $REPO
python - <<'P'
import csv,glob
f=glob.glob('gpurun_out/s8_prof_train/*/*_kernel_stats.csv')
if f:
    for i,r in enumerate(csv.DictReader(open(f[0]))):
        if i<16: print(r['Name'][:70], r['Calls'], r['TotalDurationNs'], r['Percentage'])
P
timeout -k 5 500 python tools/scope_table.py $OUT/s8_scope_table.json > $OUT/s8_scope_table.log 2> $OUT/s8_scope_table.err; echo "scope rc=$?"; cut -c1-260 $OUT/s8_scope_table.log; grep -v "MIOpen\|amdgpu.ids\|Warning\|warn" $OUT/s8_scope_table.err | tail -3
bash tools/pmc_passes.sh s8_da -- python tools/time_fb.py BL2 4 3 4 2>&1 | tail -14 | cut -c1-420
bash tools/pmc_passes.sh s8_poolREF -- python bench.py --config REF --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -12 | cut -c1-420
find $OUT -name "*.csv" -size +20M -delete
