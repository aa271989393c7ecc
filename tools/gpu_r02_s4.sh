#!/bin/bash
# Round-2 GPU session 4: two-launch-per-pass ranking chain, graph replay diagnosis, early-zero pooling variant.
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 5 120 python tools/diag_graph.py > $OUT/s4_diag_graph.log 2>&1; echo "diag rc=$?"; tail -12 $OUT/s4_diag_graph.log | cut -c1-300
timeout -k 5 400 python -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider --deselect tests/test_gpu_parity.py::test_fused_forward_is_hip_graph_capturable --deselect tests/test_gpu_parity.py::test_cached_index_build_is_graph_capturable > $OUT/s4_parity.log 2>&1
echo "parity rc=$?"; tail -12 $OUT/s4_parity.log | cut -c1-400
timeout -k 5 200 python -m pytest tests/test_gpu_conv3d.py -m gpu -q -p no:cacheprovider -k "stacks" -s > $OUT/s4_stacks.log 2>&1
echo "stacks rc=$?"; grep -E "stack training routes|passed|failed|Error" $OUT/s4_stacks.log | cut -c1-1500 | tail -5
rm -f $OUT/s4_time_rank.jsonl $OUT/s4_pool_flags.jsonl
for c in "BL2 16" "BL2 1" "REF 16" "BL5 4"; do timeout -k 5 120 python tools/time_rank.py $c 2>>$OUT/s4_time_rank.err | tail -1 | tee -a $OUT/s4_time_rank.jsonl; done
for c in "BL2 16" "BL2 16 bf16" "REF 16" "BL5 4"; do timeout -k 5 120 python tools/time_pool_flags.py $c 2>>$OUT/s4_pool_flags.err | tee -a $OUT/s4_pool_flags.jsonl; done
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s4_prof -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/s4_prof.log 2>&1
echo "rocprof rc=$?"; tail -1 $OUT/s4_prof.log | cut -c1-400
cd $REPO
python - <<'P'
import csv,glob
f=glob.glob('gpurun_out/s4_prof/*/*_kernel_stats.csv')
if f:
    for r in csv.DictReader(open(f[0])):
        print(r['Name'][:44], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
P
find $OUT -name "*.csv" -size +20M -delete
