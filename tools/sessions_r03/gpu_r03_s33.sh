#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
FBBEV_DA_BWD_PREPASS=1 timeout 600 python -m pytest tests/test_gpu_backward_projection.py -m gpu -q --timeout 600 -p no:cacheprovider -k "backward or grad or training" > $OUT/pytest_bp.log 2>&1; echo "pytest(prepass) rc=$?"; tail -2 $OUT/pytest_bp.log | cut -c1-200
rm -f $OUT/r03_exp_da_bwd_prepass.jsonl
for pp in 0 1 0 1; do
FBBEV_DA_BWD_PREPASS=$pp timeout 300 python tools/time_train.py BL2 4 4 sites 2>/dev/null > $OUT/tt_$pp.json
python - $pp <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/tt_{sys.argv[1]}.json'))
r={'FBBEV_DA_BWD_PREPASS':int(sys.argv[1]),'ms_forward_backward':d['ms_forward_backward']}
for o in d['op_sites']:
    if 'k_da_cross_attn_bwd_scatter' in o['op']: r['scatter_ms']=o['self_ms']
    if 'k_da_bwd_hitinfo' in o['op']: r['hitinfo_ms']=o['self_ms']
    if 'FusedDACrossAttentionBackward' in o['op']: r['da_backward_ms']=o['self_ms']
print(json.dumps(r)); open('gpurun_out/r03_exp_da_bwd_prepass.jsonl','a').write(json.dumps(r)+'\n')
PY
done
