#!/bin/bash
# attention block tail (output_proj + residual + LayerNorm) inside the attention kernel's workgroups: parity, S3 timing A/B
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_backward_projection.py -x -q -m gpu 2>&1 | tail -2
for t in 1 0 1 0; do
  for cfg in "BL2 4 50 4" "REF 4 50 1" "REF 1 50 1"; do
    echo "tail=$t $cfg: $(FBBEV_FUSE_ATTN_TAIL=$t python tools/time_fb.py $cfg 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_fb"],4), round(d["ms_fb_graph"],4))')"
  done
done
cd /tmp
for t in 1 0; do
  rm -rf $OUT/prof_tail; FBBEV_FUSE_ATTN_TAIL=$t timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_tail -- python $REPO/tools/time_fb.py BL2 4 10 4 > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob('$OUT/prof_tail/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('self_fused', 'attn_fused', 'rows_linear_x3<2, true')): print('tail $t', r['Name'][:50], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
done
