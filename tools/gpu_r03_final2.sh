#!/bin/bash
# Round-3 closing validation: smoke, whole GPU suite, bench (+ rocprofv3 stats and FETCH/WRITE passes), scope table, training
# step (full detector and the path alone), history step two-kernel vs one-kernel.
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
{ echo "== $(date)"; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | sort | uniq -c | head -4; nproc; grep -m1 "model name" /proc/cpuinfo; } > $OUT/box.txt 2>&1
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/box.txt; tail -1 $OUT/smoke.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/box.txt; tail -3 $OUT/pytest_gpu.log | cut -c1-200
timeout 600 python bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/box.txt; cut -c1-400 $OUT/bench.json
timeout 900 python tools/scope_table.py $OUT/r03_scope_table.json > $OUT/scope_table.log 2>&1; echo "scope rc=$?" | tee -a $OUT/box.txt
cd /tmp
rm -rf $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-storage > $OUT/prof_stats.log 2>&1; echo "rocprof stats rc=$?" | tee -a $OUT/box.txt
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-storage > $OUT/prof_fetch.log 2>&1; echo "rocprof fetch rc=$?" | tee -a $OUT/box.txt
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-storage > $OUT/prof_write.log 2>&1; echo "rocprof write rc=$?" | tee -a $OUT/box.txt
cd $REPO
python tools/pmc_to_json.py $OUT BL2_B16_tv128 "k_pool_fwd_dense2<128, 8, 4, 256, 0, false, 0, 0>" > /dev/null 2>&1; echo "pmc_to_json rc=$?" | tee -a $OUT/box.txt
timeout 300 python tools/time_fb.py BL2 4 30 4 > $OUT/fb_final_BL2_B4_L4.json 2>/dev/null; cut -c1-230 $OUT/fb_final_BL2_B4_L4.json
FBBEV_TRAIN_PROFILE=$OUT/r03_train_step_kernels_final.json timeout 900 python bench.py --mode train --steps 3 --warmup 2 > $OUT/train_final.json 2> $OUT/train_final.err; echo "train rc=$?" | tee -a $OUT/box.txt; cut -c1-250 $OUT/train_final.json
timeout 300 python tools/time_train.py BL2 4 4 > $OUT/r03_time_train_BL2_B4_L4_final.json 2>/dev/null; cat $OUT/r03_time_train_BL2_B4_L4_final.json
timeout 300 python tools/time_train.py REF 4 1 > $OUT/r03_time_train_REF_B4_final.json 2>/dev/null; cat $OUT/r03_time_train_REF_B4_final.json
rm -f $OUT/r03_time_history_fused_final.jsonl
timeout 300 python tools/time_history.py 400 400 16 1 f16 noref cbf16 vm unfused 2>/dev/null >> $OUT/r03_time_history_fused_final.jsonl
timeout 300 python tools/time_history.py 400 400 16 1 f16 noref cbf16 vm 2>/dev/null >> $OUT/r03_time_history_fused_final.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r03_time_history_fused_final.jsonl'):
    d=json.loads(l); print(d['grid'], d['history_dtype'], 'one_kernel', d['warp_conv_one_kernel'], 'step ms', d['fused_ms'])
PY
find $OUT -name "*.csv" -size +20M -delete; find $OUT -name "*kernel_trace.csv" -size +3M -delete
echo "== done $(date)" >> $OUT/box.txt
