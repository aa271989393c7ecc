#!/bin/bash
# Round-3 session 1: new full-size parity tests + whole GPU suite, scope table on HEAD, bench (with store floors),
# rocprofv3 stats of bench + of the configs[2] forward/backward projection, BEFORE-PMC of the DA sampler, micro-benchmark.
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
{ echo "== $(date)"; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | sort | uniq -c | head -4; nproc; } > $OUT/box.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/box.txt; tail -5 $OUT/pytest_gpu.log | cut -c1-400
timeout 600 python bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/box.txt; cut -c1-3000 $OUT/bench.json; tail -3 $OUT/bench.err
timeout 900 python tools/scope_table.py $OUT/r03_scope_table.json > $OUT/scope_table.log 2>&1; echo "scope rc=$?" | tee -a $OUT/box.txt; tail -30 $OUT/scope_table.log | cut -c1-260
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-storage > $OUT/prof_bench.log 2>&1; echo "rocprof bench rc=$?" | tee -a $OUT/box.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fb -- python $REPO/tools/time_fb.py BL2 4 20 4 > $OUT/prof_fb.log 2>&1; echo "rocprof fb rc=$?" | tee -a $OUT/box.txt
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_fb_before/$tag -- python $REPO/tools/time_fb.py BL2 4 5 4 > $OUT/pmc_fb_before_$tag.log 2>&1; echo "pmc $tag rc=$?" | tee -a $OUT/box.txt
done
cd $REPO
python tools/pmc_summary.py $OUT/pmc_fb_before $OUT/r03_pmc_fb_BL3_B4_before.json > /dev/null 2>&1; echo "pmc summary rc=$?" | tee -a $OUT/box.txt
test -x tools/micro/unit_sampler_pipeline || hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fb_bev_amd/csrc/hip_rt -I fb_bev_amd/csrc tools/micro/unit_sampler_pipeline.hip -o tools/micro/unit_sampler_pipeline > $OUT/usp_build.log 2>&1
for kb in 0 48 72; do timeout 120 tools/micro/unit_sampler_pipeline 160000 32 $kb >> $OUT/r03_exp_unit_sampler_pipeline.jsonl 2>&1; done
find $OUT -name "*.csv" -size +20M -delete
find $OUT -name "*kernel_trace.csv" -size +3M -delete
echo "== done $(date)" >> $OUT/box.txt
