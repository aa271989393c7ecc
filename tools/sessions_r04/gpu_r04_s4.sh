#!/bin/bash
# round 4, session 4: counters of the one-kernel DA sampler at configs[2] B=4 (the numbers VERDICT r3 items 2/3 ask for)
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/pmc_passes.sh r04_fb_fused -- python tools/time_fb.py BL2 4 5 4 > $OUT/r04_pmc_passes.log 2>&1
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04_fb_fused_pmc.json'))
for k, v in d.items():
    if 'da_cross' in k or 'rows_linear' in k or 'msda' in k:
        print(k[:40], {a: v[a] for a in sorted(v) if not a.endswith('_raw_KiB_units')})
PY
