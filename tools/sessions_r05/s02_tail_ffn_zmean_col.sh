#!/bin/bash
# round 5, session 2: the cross-attention tail + FFN as one kernel, the column-form Z-mean, per-shape constants cached: parity tests,
# A/B timings of S3 at configs[2] and at the shipped shape, bf16-storage floors in the bench line
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_block_kernels.py tests/test_gpu_backward_projection.py tests/test_gpu_parity.py -m gpu -q -s -x --timeout 900 -p no:cacheprovider > $OUT/s02_pytest.log 2>&1; echo "pytest rc=$?"
grep -o "\[observed\].*" $OUT/s02_pytest.log > $OUT/s02_observed.txt; grep "tail_ffn\|tail + FFN" $OUT/s02_observed.txt
tail -4 $OUT/s02_pytest.log | cut -c1-300
rm -f $OUT/s02_time_fb.jsonl
for rep in 1 2; do
for knobs in "" "FBBEV_FUSE_TAIL_FFN=0" "FBBEV_ZMEAN_COL=0" "FBBEV_FUSE_TAIL_FFN=0 FBBEV_ZMEAN_COL=0"; do
  for cfg in "BL2 4 40 4" "REF 1 40 1" "REF 4 40 1"; do
    env $knobs timeout 300 python tools/time_fb.py $cfg 2>/dev/null | sed "s/^{/{\"knobs\": \"$knobs\", /" >> $OUT/s02_time_fb.jsonl
  done
done
done
python - <<'PY'
import json
for l in open('gpurun_out/s02_time_fb.jsonl'):
    d = json.loads(l); print(d['knobs'] or 'default', d['config'], d['B'], d['levels'], 'eager', round(d['ms_fb'], 4), 'graph', round(d['ms_fb_graph'], 4))
PY
timeout 900 python bench.py --steps 30 --warmup 5 --no-fb-projection > $OUT/s02_bench.json 2> $OUT/s02_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/s02_bench.json').read().strip().splitlines()[-1])
r = d['roofline']
print('f32 kernel', r['kernel_ms'], 'store floor', r['store_floor_ms'], 'no gather', r['no_gather_ms'], 'no store', r.get('no_store_ms'))
print('bf16', json.dumps(d.get('bf16_storage')))
PY
cd /tmp; rm -rf $OUT/s02_prof_fb
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s02_prof_fb -- python $REPO/tools/time_fb.py BL2 4 30 4 > $OUT/s02_prof_fb.log 2>&1; echo "rocprof fb rc=$?"
cd $REPO
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/s02_prof_fb/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:24]:
    print(r['Name'][:90], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
find $OUT -name "*kernel_trace.csv" -size +20M -delete
