#!/bin/bash
# round 5, session 6: eight-point gather batches in the dense pooling kernel (the knob VERDICT r4 weak 9 left open), A/B at the shipped
# grid / BASELINE configs[0] / configs[1]; per-site device time of the path's training step at the configs[2] pyramid
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/s06_gather8.jsonl
for rep in 1 2; do
  timeout 200 python tools/time_pool_flags.py REF 16 f32 64:0x8020414 64:0x20410 64:0x8020410 >> $OUT/s06_gather8.jsonl 2>/dev/null
  timeout 200 python tools/time_pool_flags.py REF 4 f32 64:0x8020414 >> $OUT/s06_gather8.jsonl 2>/dev/null
  timeout 200 python tools/time_pool_flags.py REF 1 f32 64:0x8020414 >> $OUT/s06_gather8.jsonl 2>/dev/null
  timeout 200 python tools/time_pool_flags.py BL1 1 f32 64:0x8020414 >> $OUT/s06_gather8.jsonl 2>/dev/null
  timeout 200 python tools/time_pool_flags.py BL2 16 f32 128:0x8024424 >> $OUT/s06_gather8.jsonl 2>/dev/null
done
python - <<'PY'
import json
for l in open('gpurun_out/s06_gather8.jsonl'):
    d = json.loads(l); print(d.get('config'), d.get('B'), d.get('tv'), d.get('flags'), d.get('ms'), d.get('frac_of_8TBs'), d.get('bits_equal_first'), d.get('error', ''))
PY
timeout 900 python tools/time_train.py BL2 4 4 sites > $OUT/s06_time_train_sites.json 2>$OUT/s06_time_train.err; tail -c 300 $OUT/s06_time_train.err
python - <<'PY'
import json
txt = open('gpurun_out/s06_time_train_sites.json').read().strip().splitlines()
for l in txt[-3:]:
    try:
        d = json.loads(l)
    except Exception:
        continue
    print({k: v for k, v in d.items() if k != 'sites'})
    for s in (d.get('sites') or [])[:40]:
        print(s)
PY
