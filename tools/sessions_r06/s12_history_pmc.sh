#!/bin/bash
# round 6, session 12: counters of the history step's kernels (MFMA busy, VALU / LDS activity and waits, LDS bank conflicts, L1 traffic)
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python $REPO/tools/time_history.py 400 400 16 1 f16 noref cx3 vm"
bash tools/pmc_mfma.sh s12 -- $CMD
i=3
for set in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
           "SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  ( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/s12_mfma/p$i -- $CMD > $OUT/s12_mfma_p$i.log 2>&1 )
  echo "pmc pass $i ($set) rc=$?"
done
python tools/pmc_mfma_summary.py $OUT/s12_mfma $OUT/s12_mfma.json
find $OUT/s12_mfma -name "*.csv" -size +5M -delete
python - <<'PY'
import json
d = json.load(open('gpurun_out/s12_mfma.json'))
rows = d if isinstance(d, list) else d.get('kernels', d)
for r in rows:
    if 'k_history' in r.get('kernel', ''):
        print(json.dumps(r))
PY
