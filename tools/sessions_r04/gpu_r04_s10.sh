#!/bin/bash
# round 4, session 10: k_pool_fwd_dense_pipe (FBBEV_POOL_PIPE) vs k_pool_fwd_dense2 at the shipped grid and at BL2, runs of 2 / 4 / 8 tiles
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/r04_pool_pipe.jsonl
F1=$(python -c "from fb_bev_amd import _capi; print(hex(_capi.pool_flags(csplit=1)))")
P1=$(python -c "from fb_bev_amd import _capi; print(hex(_capi.pool_flags(csplit=1) | _capi.POOL_PIPE))")
FD=$(python -c "from fb_bev_amd import _capi; print(hex(_capi.DEFAULT_POOL_FLAGS))")
PD=$(python -c "from fb_bev_amd import _capi; print(hex(_capi.DEFAULT_POOL_FLAGS | _capi.POOL_PIPE))")
for tpw in 2 4 8; do
  for cfg in "REF 16" "REF 4" "REF 1"; do
    FBBEV_POOL_PIPE_TPW=$tpw timeout 200 python tools/time_pool_flags.py $cfg f32 64:$F1 64:$P1 128:$F1 128:$P1 128:$PD 2>/dev/null | sed "s/^{/{\"tpw\": $tpw, /" >> $OUT/r04_pool_pipe.jsonl
  done
done
FBBEV_POOL_PIPE_TPW=4 timeout 200 python tools/time_pool_flags.py BL2 16 f32 128:$FD 128:$PD 2>/dev/null | sed "s/^{/{\"tpw\": 4, /" >> $OUT/r04_pool_pipe.jsonl
FBBEV_POOL_PIPE_TPW=2 timeout 200 python tools/time_pool_flags.py BL2 16 f32 128:$FD 128:$PD 2>/dev/null | sed "s/^{/{\"tpw\": 2, /" >> $OUT/r04_pool_pipe.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r04_pool_pipe.jsonl'):
    d=json.loads(l); print('tpw', d['tpw'], d['config'], d['B'], d.get('tv'), d.get('flags'), d.get('ms'), d.get('frac_of_8TBs'), d.get('bits_equal_first'), d.get('error',''))
PY
