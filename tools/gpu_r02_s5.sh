#!/bin/bash
# Round-2 GPU session 5: look-back-free ranking chain, graph replay localisation, detector-level training-route yardstick.
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 5 100 python tools/diag_graph.py index > $OUT/s5_diag_index.log 2>&1; echo "diag index rc=$?"; grep -v amdgpu.ids $OUT/s5_diag_index.log | tail -8 | cut -c1-400
timeout -k 5 100 python tools/diag_graph.py graph > $OUT/s5_diag_graph.log 2>&1; echo "diag graph rc=$?"; grep -v amdgpu.ids $OUT/s5_diag_graph.log | tail -8 | cut -c1-400
timeout -k 5 400 python -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider --deselect tests/test_gpu_parity.py::test_fused_forward_is_hip_graph_capturable --deselect tests/test_gpu_parity.py::test_cached_index_build_is_graph_capturable > $OUT/s5_parity.log 2>&1
echo "parity rc=$?"; tail -4 $OUT/s5_parity.log | cut -c1-300
timeout -k 5 300 python -m pytest tests/test_gpu_conv3d.py -m gpu -q -p no:cacheprovider -k "stacks or training_route_equals" -s > $OUT/s5_routes.log 2>&1
echo "routes rc=$?"; grep -E "stack training routes|loss mfma|per block|passed|failed|Error" $OUT/s5_routes.log | cut -c1-1800 | tail -8
rm -f $OUT/s5_time_rank.jsonl
for c in "BL2 16" "BL2 1" "REF 16" "BL5 4"; do timeout -k 5 120 python tools/time_rank.py $c 2>>$OUT/s5_time_rank.err | tail -1 | tee -a $OUT/s5_time_rank.jsonl; done
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s5_prof -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/s5_prof.log 2>&1
echo "rocprof rc=$?"; tail -1 $OUT/s5_prof.log | cut -c1-300
cd $REPO
python - <<'P'
import csv,glob
f=glob.glob('gpurun_out/s5_prof/*/*_kernel_stats.csv')
if f:
    for r in csv.DictReader(open(f[0])):
        print(r['Name'][:44], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
P
find $OUT -name "*.csv" -size +20M -delete
