#!/bin/bash
# the whole GPU suite + smoke + bench at HEAD
REPO=$(pwd); OUT=$REPO/gpurun_out/s07; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/s07/bench.json'))
print('value', d['value'], 'frac', d['roofline']['frac'])
for k in ('fb_projection','fb_projection_train'):
    f=d.get(k,{})
    print(k, {kk:f.get(kk) for kk in ('error','value','ms_per_step','fp32_gemm_route_ms','forward_ms_train_mode','da_backward_ms_hip_events','launches_per_step')})
PY
