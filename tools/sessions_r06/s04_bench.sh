#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/s04; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/s04/bench.json'))
print('value', d['value'], 'frac', d['roofline']['frac'])
for k in ('fb_projection','fb_projection_train'):
    f=d.get(k,{})
    print(k, {kk:f.get(kk) for kk in ('error','value','ms_per_step','step_gpu_ms_p10_p50_p90','fp32_gemm_route_ms','forward_ms_train_mode','da_backward_ms_hip_events','kernel_ms_one_step_profile','launches_per_step','gpu_over_cpu')})
    print('   cpu', f.get('cpu_baseline'))
    print('   roof', {kk:vv for kk,vv in (f.get('roofline') or {}).items() if kk in ('kernel','achieved','frac','kernel_ms','algorithmic_bytes_per_launch')})
PY
