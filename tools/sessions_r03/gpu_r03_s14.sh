#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/pmc_passes.sh r03_hist_fused -- python tools/time_history.py 400 400 16 1 f16 noref cbf16 vm 2>&1 | tail -8
bash tools/pmc_passes.sh r03_hist_unfused -- python tools/time_history.py 400 400 16 1 f16 noref cbf16 vm unfused 2>&1 | tail -3
python - <<'PY'
import json
for tag in ('fused','unfused'):
    d=json.load(open(f'gpurun_out/r03_hist_{tag}_pmc.json'))
    for k,v in d.items():
        if 'k_history' in k and 'conv_t' not in k:
            print(tag, k[:40], json.dumps({a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()}))
PY
