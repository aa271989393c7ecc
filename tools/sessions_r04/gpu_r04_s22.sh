#!/bin/bash
# round 4, session 22: Z planes of the Z-mean dealt to their own workgroups (fbbev_pool_zmean_split): tests + S3 at the shipped shape
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward_projection.py tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "zmean or write_once or module_vs_oracle or graphed or full_size or fused_module" 2>&1 | tail -2
rm -f $OUT/r04_zmean_zsplit.jsonl
for zg in 0 0; do
  for cfg in "REF 1 1" "REF 4 1" "BL2 4 4" "BL2 1 4"; do
    set -- $cfg
    FBBEV_ZMEAN_ZGROUPS=$zg timeout 300 python tools/time_fb.py $1 $2 50 $3 2>/dev/null | sed "s/^{/{\"zmean_zgroups\": $zg, /" >> $OUT/r04_zmean_zsplit.jsonl
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r04_zmean_zsplit.jsonl'):
    d=json.loads(l); print('zmean_zgroups(0=heuristic,1=single pass)', d['zmean_zgroups'], d['config'], d['B'], 'fb', round(d['ms_fb'],4), 'graph', round(d['ms_fb_graph'],4))
PY
