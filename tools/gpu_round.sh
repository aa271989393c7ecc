#!/bin/bash
# One GPU-box validation session: smoke, the whole GPU suite, bench (f32 + bf16 storage), rocprofv3 kernel stats and the
# FETCH_SIZE / WRITE_SIZE passes behind roofline.traffic.       gpurun --timeout 1500 -- 'bash tools/gpu_round.sh [quick]'
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
MODE=${1:-full}
{ echo "== $(date) mode=$MODE"; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | sort | uniq -c | head -8; nproc; grep -m1 "model name" /proc/cpuinfo; } > $OUT/box.txt 2>&1
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/box.txt; tail -2 $OUT/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -s --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/box.txt; tail -4 $OUT/pytest_gpu.log | cut -c1-300
timeout 900 python bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/box.txt; cut -c1-2200 $OUT/bench.json; tail -3 $OUT/bench.err
if [ "$MODE" != "quick" ]; then
  timeout 300 python bench.py --steps 30 --warmup 5 --storage bf16 --no-cpu-baseline > $OUT/bench_bf16.json 2>> $OUT/bench.err; cut -c1-300 $OUT/bench_bf16.json
  cd /tmp
  rm -rf $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-storage --no-fb-projection > $OUT/prof_stats.log 2>&1; echo "rocprof stats rc=$?" | tee -a $OUT/box.txt
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-storage --no-fb-projection > $OUT/prof_fetch.log 2>&1; echo "rocprof fetch rc=$?" | tee -a $OUT/box.txt
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-storage --no-fb-projection > $OUT/prof_write.log 2>&1; echo "rocprof write rc=$?" | tee -a $OUT/box.txt
  cd $REPO
  python tools/pmc_to_json.py $OUT BL2_B16_tv128 "k_pool_fwd_dense2<128, 8, 4, 256, 0, false, 0, 0"
  find $OUT -name "*.csv" -size +20M -delete
fi
echo "== done $(date)" >> $OUT/box.txt
