#!/bin/bash
# round 4, session 15: the scope table (S1-S6) of HEAD, training-path steps, PMC passes of the S3 scope
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python tools/scope_table.py $OUT/r04_scope_table.json > $OUT/r04_scope_table.log 2>&1; echo "scope rc=$?"; tail -3 $OUT/r04_scope_table.log | cut -c1-300
rm -f $OUT/r04_time_train.jsonl
timeout 300 python tools/time_train.py BL2 4 4 >> $OUT/r04_time_train.jsonl 2>/dev/null
timeout 300 python tools/time_train.py REF 4 1 >> $OUT/r04_time_train.jsonl 2>/dev/null
cut -c1-500 $OUT/r04_time_train.jsonl
bash tools/pmc_passes.sh r04_fb_final -- python tools/time_fb.py BL2 4 5 4 > $OUT/r04_pmc_final.log 2>&1
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04_fb_final_pmc.json'))
for k, v in d.items():
    if 'fused' in k or 'rows_linear_x3<' in k:
        print(k[:44], {a: v[a] for a in ('launches','TCP_TOTAL_CACHE_ACCESSES_sum','WRITE_SIZE_bytes','FETCH_SIZE','SQ_INSTS_VALU','frac_parked_waitcnt_barrier','frac_issue_stall','frac_issuing','L2_hit_rate') if a in v})
PY
