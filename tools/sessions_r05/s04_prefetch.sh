#!/bin/bash
# round 5, session 4: side-stream prefetch of the Z-mean-independent part of the backward projection; per-sample scale of the
# owned-plane scatter; tests + A/B timing
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_block_kernels.py tests/test_gpu_backward_projection.py tests/test_gpu_full_model.py -m gpu -q -s -x --timeout 900 -p no:cacheprovider > $OUT/s04_pytest.log 2>&1; echo "pytest rc=$?"
grep -o "\[observed\].*" $OUT/s04_pytest.log | grep -i "prefetch\|outlier"
tail -3 $OUT/s04_pytest.log | cut -c1-300
grep -E "^E  |FAILED" $OUT/s04_pytest.log | head
rm -f $OUT/s04_time_fb.jsonl
for rep in 1 2; do
for knobs in "" "FBBEV_BP_PREFETCH=0"; do
  for cfg in "BL2 4 40 4" "REF 1 40 1" "REF 4 40 1" "BL2 1 40 4"; do
    env $knobs timeout 300 python tools/time_fb.py $cfg 2>/dev/null | sed "s/^{/{\"knobs\": \"$knobs\", /" >> $OUT/s04_time_fb.jsonl
  done
done
done
python - <<'PY'
import json
for l in open('gpurun_out/s04_time_fb.jsonl'):
    d = json.loads(l); print(d['knobs'] or 'default', d['config'], d['B'], d['levels'], 'eager', round(d['ms_fb'], 4), 'graph', round(d['ms_fb_graph'], 4))
PY
timeout 600 python tools/time_train.py BL2 4 4 > $OUT/s04_time_train.json 2>$OUT/s04_time_train.err; tail -c 600 $OUT/s04_time_train.json
