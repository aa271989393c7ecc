#!/bin/bash
# round 5, session 10: 16-bit camera tokens as head planes on the one-kernel DA sampler (fbbev_rows_linear_x3_planes_e +
# fbbev_da_cross_attn_fused_e): kernel-level test, module tests, A/B timing against fp32 tokens and the round-3 16-bit route
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_block_kernels.py tests/test_gpu_backward_projection.py -m gpu -q -s -x --timeout 900 -p no:cacheprovider > $OUT/s10_pytest.log 2>&1; echo "pytest rc=$?"
grep -o "\[observed\].*" $OUT/s10_pytest.log | grep -i "16\|head planes"
tail -3 $OUT/s10_pytest.log | cut -c1-300
grep -E "^E  |FAILED" $OUT/s10_pytest.log | head
rm -f $OUT/s10_time_fb.jsonl
for rep in 1 2; do
for knobs in "f32" "bf16" "f16" "bf16 FBBEV_DA_16BIT_PLANES=0"; do
  set -- $knobs; dt=$1; shift
  for cfg in "BL2 4 40 4" "REF 1 40 1"; do
    env $@ timeout 300 python tools/time_fb.py $cfg $dt 2>/dev/null | sed "s/^{/{\"knobs\": \"$knobs\", /" >> $OUT/s10_time_fb.jsonl
  done
done
done
python - <<'PY'
import json
for l in open('gpurun_out/s10_time_fb.jsonl'):
    d = json.loads(l); print(d['knobs'] or 'default', d['config'], d['B'], d['levels'], 'eager', round(d['ms_fb'], 4), 'graph', round(d['ms_fb_graph'], 4))
PY
# kernel durations of the 16-bit route
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s10_prof -- python $REPO/tools/time_fb.py BL2 4 20 4 bf16 > $OUT/s10_prof.log 2>&1; echo "rocprof rc=$?"
cd $REPO
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/s10_prof/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:10]:
        print(r['Name'][:90], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), r['Percentage'])
PY
