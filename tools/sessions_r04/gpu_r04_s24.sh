#!/bin/bash
# round 4, session 24: tile / channel-group / XCD-chunk sweep of the dense pooling kernel at SMALL batches (BL2 B = 4 and 1, REF B = 4 and 1):
# the defaults were tuned at B = 16 only
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for cfg in "BL2 4" "BL2 1" "REF 4" "REF 1"; do
  set -- $cfg
  timeout 400 python tools/sweep_pool.py $1 $2 f32 > $OUT/r04_sweep_pool_$1_B$2.jsonl 2>/dev/null
  python - "$OUT/r04_sweep_pool_$1_B$2.jsonl" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
hdr, rows = rows[0], rows[1:]
rows.sort(key=lambda r: r['ms'])
print(hdr['config'], hdr['B'], 'best:', [(r['variant'], r['ms'], r['frac']) for r in rows[:4]], 'bits_ok_all', all(r['bits_ok'] for r in rows))
cur = [r for r in rows if r['variant'] in ('tv128_cs2_wg256_swz4_cpl8', 'tv64_cs1_wg256_swz4_cpl8')]
print('   current defaults:', [(r['variant'], r['ms']) for r in cur])
PY
done
