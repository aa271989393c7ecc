#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/pmc_passes.sh r03_fb_final -- python tools/time_fb.py BL2 4 5 4 2>&1 | tail -8
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_fb_final_pmc.json'))
keep={k:v for k,v in d.items() if k.startswith('k_') or 'k_' in k[:12]}
json.dump(keep, open('gpurun_out/r03_pmc_fb_BL3_B4_final.json','w'), indent=1)
for k,v in keep.items():
    if 'rows_linear' in k or 'fwd_pipe' in k or 'msda_fwd_unit' in k:
        print(k[:50], {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ('launches','frac_parked_waitcnt_barrier','frac_issue_stall','frac_issuing','L2_hit_rate','FETCH_SIZE_bytes_raw_KiB_units','WRITE_SIZE_bytes','SQ_INSTS_VALU','TCP_TOTAL_CACHE_ACCESSES_sum','SQ_LDS_BANK_CONFLICT','SQ_BUSY_CYCLES')})
PY
