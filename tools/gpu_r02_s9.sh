#!/bin/bash
# Round-2 GPU session 9: configs[3] test, training step with the separable trilinear upsample.
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 5 400 python -m pytest tests/test_gpu_full_model.py -m gpu -q -s -p no:cacheprovider > $OUT/s9_tests.log 2>&1
echo "tests rc=$?"; grep -E "configs\[|passed|failed|Error|^E  " $OUT/s9_tests.log | cut -c1-900 | tail -8
timeout -k 5 400 python bench.py --mode train --steps 5 --warmup 2 > $OUT/s9_bench_train.json 2> $OUT/s9_bench_train.err; echo "bench train rc=$?"; cut -c1-330 $OUT/s9_bench_train.json; grep -v "MIOpen\|amdgpu.ids" $OUT/s9_bench_train.err | tail -3
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s9_prof_train -- python $REPO/bench.py --mode train --steps 3 --warmup 2 > $OUT/s9_prof_train.log 2>&1
echo "rocprof train rc=$?"
cd $REPO
python - <<'P'
import csv,glob
f=glob.glob('gpurun_out/s9_prof_train/*/*_kernel_stats.csv')
if f:
    for i,r in enumerate(csv.DictReader(open(f[0]))):
        if i<22: print(r['Name'][:80], r['Calls'], r['TotalDurationNs'], r['Percentage'])
P
find $OUT -name "*.csv" -size +20M -delete
