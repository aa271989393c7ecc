#!/bin/bash
# fused DA: 4 heads per workgroup (two 256-thread workgroups per patch and CU) against 8 (one 512-thread workgroup)
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_backward_projection.py -x -q -m gpu 2>&1 | tail -2
for hw in 8 4; do
  for cfg in "BL2 4 50 4" "REF 4 50 1" "REF 1 50 1"; do
    echo "hw=$hw $cfg: $(FBBEV_DA_FUSED_HW=$hw python tools/time_fb.py $cfg 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_fb"],4), round(d["ms_fb_graph"],4))')"
  done
done
cd /tmp
for hw in 8 4; do
  rm -rf $OUT/prof_hw; FBBEV_DA_FUSED_HW=$hw timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_hw -- python $REPO/tools/time_fb.py BL2 4 10 4 > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob('$OUT/prof_hw/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'da_cross_attn_fused' in r['Name']: print('hw $hw', r['Name'][:44], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
done
