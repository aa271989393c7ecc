#!/bin/bash
# round 5, closing validation after the latency-chain work (sessions 10-14): smoke, the whole GPU suite (observed errors collected), bench.py (forward + bf16 storage + index cache +
# fb_projection legs), rocprofv3 kernel stats of the bench command and of S3 at configs[2], FETCH_SIZE / WRITE_SIZE passes, counters
# of the S3 kernels, scope table, training step of the path and of the detector
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/gpu_round.sh full
grep -o "\[observed\].*" $OUT/pytest_gpu.log > $OUT/r05_gpu_tests_observed_final.txt; wc -l $OUT/r05_gpu_tests_observed_final.txt
cd /tmp; rm -rf $OUT/s15_prof_fb
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s15_prof_fb -- python $REPO/tools/time_fb.py BL2 4 30 4 > $OUT/s15_prof_fb.log 2>&1; echo "rocprof fb rc=$?"
cd $REPO
tail -1 $OUT/s15_prof_fb.log | cut -c1-400
bash tools/pmc_passes.sh s15_fb -- python tools/time_fb.py BL2 4 5 4 > $OUT/s15_pmc_fb.log 2>&1; tail -3 $OUT/s15_pmc_fb.log | cut -c1-300
timeout 1500 python tools/scope_table.py $OUT/r05_scope_table.json > $OUT/s15_scope_table.log 2>&1; echo "scope table rc=$?"; tail -32 $OUT/s15_scope_table.log | cut -c1-250
timeout 600 python tools/time_train.py BL2 4 4 > $OUT/s15_time_train.json 2>/dev/null; cut -c1-400 $OUT/s15_time_train.json
timeout 900 python bench.py --mode train --steps 6 --warmup 3 > $OUT/s15_bench_train.json 2>$OUT/s15_bench_train.err; echo "bench train rc=$?"; cut -c1-1500 $OUT/s15_bench_train.json
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*.csv" -size +30M -delete
