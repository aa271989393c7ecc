#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python tools/scope_table.py $OUT/r03_scope_s6_variants.json only_s6 > $OUT/scope_s6.log 2>&1; echo "scope rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_scope_s6_variants.json'))
for r in d['rows']: print(r['history_convs'], r['ring_layout'], r['ms_p10_p50_p90'], round(r['samples_per_s'],1))
PY
timeout 900 python -m pytest tests/test_gpu_history.py tests/test_gpu_full_model.py -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_hist.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_hist.log | cut -c1-300
