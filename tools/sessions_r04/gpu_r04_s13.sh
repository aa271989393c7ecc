#!/bin/bash
# round 4, session 13: output_proj / FFN tail + residual + LayerNorm in one kernel (fbbev_rows_linear_x3_ln): tests + S3 on / off
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward_projection.py tests/test_gpu_full_model.py -m gpu -q -s -x --timeout 600 -p no:cacheprovider > $OUT/r04_s13_pytest.log 2>&1; echo "pytest rc=$?"
grep -E "BackwardProjection full size|passed|failed|Error|error" $OUT/r04_s13_pytest.log | cut -c1-300 | tail -8
rm -f $OUT/r04_time_fb_ln.jsonl
for f in 1 0 1 0; do
  FBBEV_FUSE_OUT_NORM=$f timeout 300 python tools/time_fb.py BL2 4 50 4 2>/dev/null | sed "s/^{/{\"fuse_out_norm\": $f, /" >> $OUT/r04_time_fb_ln.jsonl
  FBBEV_FUSE_OUT_NORM=$f timeout 300 python tools/time_fb.py REF 4 50 1 2>/dev/null | sed "s/^{/{\"fuse_out_norm\": $f, /" >> $OUT/r04_time_fb_ln.jsonl
  FBBEV_FUSE_OUT_NORM=$f timeout 300 python tools/time_fb.py REF 1 50 1 2>/dev/null | sed "s/^{/{\"fuse_out_norm\": $f, /" >> $OUT/r04_time_fb_ln.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r04_time_fb_ln.jsonl'):
    d=json.loads(l); print('fuse_out_norm', d['fuse_out_norm'], d['config'], d['B'], 'L', d['levels'], 'fb', round(d['ms_fb'],4), 'graph', round(d['ms_fb_graph'],4))
PY
