#!/bin/bash
# round 5, closing session (after the last code change: Z-mean metadata prefetch reverted, bench legs fixed): smoke, GPU suite,
# bench, rocprofv3 stats of the forward legs and of S3, FETCH / WRITE passes, counters of the S3 kernels
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/gpu_round.sh full
grep -o "\[observed\].*" $OUT/pytest_gpu.log > $OUT/r05_gpu_tests_observed_final.txt; wc -l $OUT/r05_gpu_tests_observed_final.txt
cd /tmp; rm -rf $OUT/s08_prof_fb
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s08_prof_fb -- python $REPO/tools/time_fb.py BL2 4 30 4 > $OUT/s08_prof_fb.log 2>&1; echo "rocprof fb rc=$?"
cd $REPO
tail -1 $OUT/s08_prof_fb.log | cut -c1-400
rm -rf $OUT/s08_fb_pmc
bash tools/pmc_passes.sh s08_fb -- python tools/time_fb.py BL2 4 5 4 > $OUT/s08_pmc_fb.log 2>&1; tail -2 $OUT/s08_pmc_fb.log | cut -c1-300
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*.csv" -size +30M -delete
