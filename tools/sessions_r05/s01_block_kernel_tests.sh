#!/bin/bash
# round 5, session 1: kernel-level GPU tests of the round-4 default kernels + the one-call lift-splat entry +
# TRTBEVPoolv2, observed errors collected; bench.py with the fb_projection leg; rocprofv3 baseline of S3 at configs[2]
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
{ echo "== $(date)"; rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit" | sort | uniq -c | head -4; nproc; grep -m1 "model name" /proc/cpuinfo; } > $OUT/s01_box.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_block_kernels.py tests/test_gpu_bevdet.py "tests/test_gpu_parity.py::test_lift_splat_fused_one_entry_equals_the_four_call_sequence" \
   -m gpu -q -s --timeout 900 -p no:cacheprovider > $OUT/s01_pytest_new.log 2>&1; echo "pytest(new) rc=$?" | tee -a $OUT/s01_box.txt
grep -o "\[observed\].*" $OUT/s01_pytest_new.log > $OUT/r05_gpu_tests_observed.txt; wc -l $OUT/r05_gpu_tests_observed.txt
tail -5 $OUT/s01_pytest_new.log | cut -c1-400
grep -E "FAILED|Error|assert" $OUT/s01_pytest_new.log | head -20 | cut -c1-300
timeout 900 python bench.py --steps 30 --warmup 5 > $OUT/s01_bench.json 2> $OUT/s01_bench.err; echo "bench rc=$?" | tee -a $OUT/s01_box.txt
python - <<'PY'
import json
d = json.loads(open('gpurun_out/s01_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', d['roofline']['frac'], 'bf16', d.get('bf16_storage', {}).get('roofline_frac'))
print(json.dumps(d.get('fb_projection'), indent=1)[:3000])
PY
tail -3 $OUT/s01_bench.err
cd /tmp; rm -rf $OUT/s01_prof_fb
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s01_prof_fb -- python $REPO/tools/time_fb.py BL2 4 30 4 > $OUT/s01_prof_fb.log 2>&1; echo "rocprof fb rc=$?" | tee -a $OUT/s01_box.txt
cd $REPO
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/s01_prof_fb/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:28]:
    print(r['Name'][:90], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
tail -2 $OUT/s01_prof_fb.log | cut -c1-600
find $OUT -name "*kernel_trace.csv" -size +20M -delete
echo "== done $(date)" >> $OUT/s01_box.txt
