#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for kb in 68 100 144; do
FBBEV_DA_BWD_LDS_KB=$kb timeout 300 python tools/time_train.py BL2 4 4 sites > $OUT/r03_time_train_da_lds_$kb.json 2>$OUT/time_train.err; tail -1 $OUT/time_train.err
python - $kb <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r03_time_train_da_lds_{sys.argv[1]}.json'))
print('DA_BWD_LDS_KB', sys.argv[1], d['ms_forward_backward'])
for r in d['op_sites'][:40]:
    if 'da_cross_attn_bwd' in r['op'] or 'FusedDACrossAttentionBackward' in r['op'] or 'da_bwd_reduce' in r['op']: print(round(r['self_ms'],3), r['calls'], r['op'][:60])
PY
FBBEV_DA_BWD_LDS_KB=$kb timeout 300 python tools/time_train.py REF 4 1 2>/dev/null | cut -c1-200
done
