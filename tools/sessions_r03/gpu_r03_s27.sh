#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -2 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench.json'))
print('value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'gpu p50', d['step_gpu_ms_p10_p50_p90'][1], 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],4))
print(d['launch'][:60], d['graph_step_equals_eager_step'], d['launch_probe'])
print(json.dumps(d['cpu_baseline'])[:200])
PY
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-alt-storage 2>/dev/null | cut -c1-200
