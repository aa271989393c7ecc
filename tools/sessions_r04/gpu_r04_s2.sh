#!/bin/bash
# round 4, session 2: head-plane layout micro-benchmark (tools/micro/plane_sampler.hip) at the levels of the configs[2] pyramid,
# depth net after the 1x1-upsample broadcast.
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/r04_exp_plane_sampler.jsonl
for lvl in "32 88" "16 44" "8 22"; do
  for coh in 1 0; do
    timeout 120 tools/micro/plane_sampler 160000 32 $coh $lvl >> $OUT/r04_exp_plane_sampler.jsonl
  done
done
timeout 120 tools/micro/plane_sampler 40000 32 1 16 44 >> $OUT/r04_exp_plane_sampler.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r04_exp_plane_sampler.jsonl'):
    d = json.loads(l)
    print(d['Q'], d['level'], 'coh', d['coherent_offsets'], {k[:-3]: v for k, v in d.items() if k.endswith('_ms')}, d['max_abs_diff_vs_rows'])
PY
timeout 300 python tools/time_depthnet.py 4 > $OUT/r04_time_depthnet.json 2>/dev/null; cat $OUT/r04_time_depthnet.json
timeout 300 python tools/time_depthnet.py 1 >> $OUT/r04_time_depthnet.json 2>/dev/null; tail -1 $OUT/r04_time_depthnet.json
