#!/bin/bash
# Round-3 validation session: smoke, whole GPU suite, bench (f32 + bf16 storage, store floors), rocprofv3 stats + FETCH/WRITE
# passes of the bench, scope table, forward+backward projection stats + PMC of the DA sampler, training step, tolerance mode.
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
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fb -- python $REPO/tools/time_fb.py BL2 4 20 4 > $OUT/prof_fb.log 2>&1; echo "rocprof fb rc=$?" | tee -a $OUT/box.txt
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_fb_after/$tag -- python $REPO/tools/time_fb.py BL2 4 5 4 > $OUT/pmc_fb_after_$tag.log 2>&1; echo "pmc $tag rc=$?" | tee -a $OUT/box.txt
done
cd $REPO
python tools/pmc_to_json.py $OUT BL2_B16_tv128 "k_pool_fwd_dense2<128, 8, 4, 256, 0, false, 0, 0>" > /dev/null 2>&1; echo "pmc_to_json rc=$?" | tee -a $OUT/box.txt
python tools/pmc_summary.py $OUT/pmc_fb_after $OUT/r03_pmc_fb_BL3_B4_after.json > /dev/null 2>&1
for w in 2 3; do FBBEV_DA_PIPE_WPS=$w timeout 300 python tools/time_fb.py BL2 4 30 4 > $OUT/fb_final_wps$w.json 2>/dev/null; echo "wps=$w $(cut -c1-230 $OUT/fb_final_wps$w.json)"; done
timeout 300 python tools/time_fb.py REF 1 30 1 > $OUT/fb_REF_B1.json 2>/dev/null; cut -c1-330 $OUT/fb_REF_B1.json
FBBEV_TRAIN_PROFILE=$OUT/r03_train_step_kernels_final.json timeout 900 python bench.py --mode train --steps 3 --warmup 2 > $OUT/train_final.json 2> $OUT/train_final.err; echo "train rc=$?" | tee -a $OUT/box.txt; cut -c1-250 $OUT/train_final.json
timeout 900 python bench.py --mode train --steps 3 --warmup 2 --conv-dtype f32 > $OUT/train_final_f32.json 2>/dev/null; cut -c1-250 $OUT/train_final_f32.json
T=0x2000000
rm -f $OUT/r03_pool_tolerance_mode.jsonl
python tools/time_pool_flags.py REF 16 f32 64:$(printf "0x%x" $((0x24414 | T))) >> $OUT/r03_pool_tolerance_mode.jsonl 2>/dev/null
python tools/time_pool_flags.py BL1 4 f32 64:$(printf "0x%x" $((0x24414 | T))) 128:0x24424 128:$(printf "0x%x" $((0x24424 | T))) >> $OUT/r03_pool_tolerance_mode.jsonl 2>/dev/null
python tools/time_pool_flags.py BL2 16 f32 128:$(printf "0x%x" $((0x24424 | T))) >> $OUT/r03_pool_tolerance_mode.jsonl 2>/dev/null
cat $OUT/r03_pool_tolerance_mode.jsonl | cut -c1-170
for i in 1 2 3; do timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt-storage --streams 2 --pipeline graphs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'value': d['value'], 'pipelined': d['pipelined']['value'], 'same': d['pipelined']['volumes_identical_to_single_stream']}))" >> $OUT/r03_exp_stream_overlap.jsonl; done; cat $OUT/r03_exp_stream_overlap.jsonl
find $OUT -name "*.csv" -size +20M -delete; find $OUT -name "*kernel_trace.csv" -size +3M -delete
echo "== done $(date)" >> $OUT/box.txt
