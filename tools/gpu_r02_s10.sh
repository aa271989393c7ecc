#!/bin/bash
# Round-2 GPU session 10: 16-bit pooling with a 16-bit LDS tile, history timings at the configs[4] grid, full suite, final numbers.
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/s10_pytest_gpu.log 2>&1
echo "pytest gpu rc=$?"; tail -3 $OUT/s10_pytest_gpu.log | cut -c1-300
rm -f $OUT/s10_pool16.jsonl $OUT/s10_hist.jsonl
for st in bf16 f16; do timeout -k 5 120 python tools/time_pool_flags.py BL2 16 $st 128:0x24424 256:0x24424 512:0x24424 64:0x24414 2>>$OUT/s10_pool16.err | tee -a $OUT/s10_pool16.jsonl; done
timeout -k 5 120 python tools/time_pool_flags.py BL5 4 f16 128:0x24424 256:0x24424 512:0x24424 2>>$OUT/s10_pool16.err | tee -a $OUT/s10_pool16.jsonl
for st in f32 bf16; do timeout -k 5 200 python bench.py --steps 30 --warmup 5 --storage $st --no-cpu-baseline 2>>$OUT/s10_bench.err | tee $OUT/s10_bench_$st.json | cut -c1-200; python - <<P
import json; d=json.load(open('$OUT/s10_bench_$st.json')); print('$st', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['tile_voxels'])
P
done
for a in "100 100 8 1 f32" "100 100 8 1 f16 noref" "400 400 16 1 f16 noref" "400 400 16 1 f32 noref" "200 200 16 4 f16 noref"; do timeout -k 5 200 python tools/time_history.py $a 2>>$OUT/s10_hist.err | tail -1 | tee -a $OUT/s10_hist.jsonl; done
timeout -k 5 500 python tools/scope_table.py $OUT/s10_scope_table.json > $OUT/s10_scope_table.log 2> $OUT/s10_scope_table.err; echo "scope rc=$?"; grep -E "S5|S4" $OUT/s10_scope_table.log | cut -c1-230
bash tools/pmc_passes.sh s10_bf16 -- python bench.py --storage bf16 --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | grep -E "dense2|rc=" | cut -c1-400
find $OUT -name "*.csv" -size +20M -delete
