#!/bin/bash
# Round-6 validation + evidence session on one GPU box: smoke, the whole GPU suite (observed values kept), bench (default line),
# rocprofv3 kernel stats of the headline / the S3 step / the path's training step, the FETCH / WRITE passes behind roofline.traffic,
# PMC passes (issue / LDS / L1 / L2 counters, MFMA busy) over S3 and the training step, the scope table and the pool-backward rows.
#     gpurun --timeout 3000 -- 'bash tools/gpu_round6.sh [quick]'          -> gpurun_out/r06/  (copy what is cited into profiles/r06_final/)
REPO=$(pwd); OUT=$REPO/gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
MODE=${1:-full}
{ echo "== $(date) mode=$MODE"; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | sort | uniq -c | head -8; nproc; grep -m1 "model name" /proc/cpuinfo; git -C $REPO rev-parse --short HEAD 2>/dev/null; } > $OUT/box.txt 2>&1
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/box.txt; tail -1 $OUT/smoke.log
timeout 1800 python -m pytest tests -m gpu -q -s --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/box.txt; tail -3 $OUT/pytest_gpu.log | cut -c1-300
grep -E "max\|err|max\|coor|vs oracle|vs fp64|worst|deviation|observed|forward:|get_lidar|bits changed|BackwardProjection full" $OUT/pytest_gpu.log | cut -c1-400 > $OUT/gpu_tests_observed.txt
tail -5 $OUT/pytest_gpu.log > $OUT/pytest_gpu_tail.txt
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/box.txt; cut -c1-600 $OUT/bench.json; tail -2 $OUT/bench.err
python tools/time_rows_kernels.py > $OUT/time_rows_kernels.jsonl 2>/dev/null
python tools/time_bwd.py BL2 16 > $OUT/time_bwd_BL2_B16.jsonl 2>/dev/null; python tools/time_bwd.py BL2 4 >> $OUT/time_bwd_BL2_B16.jsonl 2>/dev/null
python tools/train_path.py BL2 4 4 --steps 30 --checksum --sites > $OUT/time_train_path_BL2_B4_L4.json 2>/dev/null
FBBEV_TRAIN_FUSED=0 python tools/train_path.py BL2 4 4 --steps 20 > $OUT/time_train_path_composite_route.json 2>/dev/null
python tools/train_path.py REF 4 1 --steps 30 > $OUT/time_train_path_REF_B4.json 2>/dev/null
cd /tmp
rm -rf $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write $OUT/prof_fb $OUT/prof_train $OUT/prof_bwd
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-storage --no-fb-projection --no-reference-gpu --streams 1 > $OUT/prof_stats.log 2>&1; echo "rocprof headline rc=$?" | tee -a $OUT/box.txt
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-storage --no-fb-projection --no-reference-gpu --streams 1 > $OUT/prof_fetch.log 2>&1; echo "rocprof fetch rc=$?" | tee -a $OUT/box.txt
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-storage --no-fb-projection --no-reference-gpu --streams 1 > $OUT/prof_write.log 2>&1; echo "rocprof write rc=$?" | tee -a $OUT/box.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fb -- python $REPO/tools/time_fb.py BL2 4 20 4 > $OUT/prof_fb.log 2>&1; echo "rocprof S3 rc=$?" | tee -a $OUT/box.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -- python $REPO/tools/train_path.py BL2 4 4 --profile-steps 10 > $OUT/prof_train.log 2>&1; echo "rocprof train rc=$?" | tee -a $OUT/box.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bwd -- python $REPO/tools/time_bwd.py BL2 16 > $OUT/prof_bwd.log 2>&1; echo "rocprof pool bwd rc=$?" | tee -a $OUT/box.txt
cd $REPO
python tools/pmc_to_json.py $OUT BL2_B16_tv128 "k_pool_fwd_dense2<128, 8, 4, 256, 0, false, 0, 0"
for d in prof_stats prof_fb prof_train prof_bwd; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/rocprofv3_$d.csv; done
if [ "$MODE" != "quick" ]; then
  bash tools/pmc_passes.sh r06/fb_BL3_B4 -- python tools/time_fb.py BL2 4 5 4 > $OUT/pmc_fb.log 2>&1
  bash tools/pmc_passes.sh r06/train_path -- python tools/train_path.py BL2 4 4 --profile-steps 3 > $OUT/pmc_train.log 2>&1
  bash tools/pmc_mfma.sh r06/fb_BL3_B4 -- python tools/time_fb.py BL2 4 5 4 > $OUT/pmc_mfma_fb.log 2>&1
  bash tools/pmc_mfma.sh r06/train_path -- python tools/train_path.py BL2 4 4 --profile-steps 3 > $OUT/pmc_mfma_train.log 2>&1
  bash tools/pmc_mfma.sh r06/hist -- python tools/time_history.py 400 400 16 1 f16 noref cx3 vm > $OUT/pmc_mfma_hist.log 2>&1
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_hist -- python $REPO/tools/time_history.py 400 400 16 1 f16 noref cx3 vm > $OUT/prof_hist.log 2>&1 ); f=$(find $OUT/prof_hist -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/rocprofv3_prof_hist.csv
  for i in 1 2; do python tools/time_history.py 400 400 16 1 f16 noref cx3 vm >> $OUT/time_history_x3_vm.jsonl 2>/dev/null; done
  timeout 1500 python tools/scope_table.py $OUT/scope_table.json > $OUT/scope_table.log 2>&1; echo "scope table rc=$?" | tee -a $OUT/box.txt
fi
find $OUT -name "*.csv" -size +20M -delete
find $OUT -name "*kernel_trace.csv" -size +3M -delete
echo "== done $(date)" >> $OUT/box.txt
du -sh $OUT
