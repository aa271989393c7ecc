#!/bin/bash
# round 6, session 19: counters of the one-kernel history step (k_history_fused_x3) next to the warp kernel's
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
export HIST_FUSED_X3=1
CMD="python $REPO/tools/time_history.py 400 400 16 1 f16 noref cx3 vm"
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  ( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/s19b_mfma/p$i -- $CMD > $OUT/s19b_mfma_p$i.log 2>&1 )
  echo "pmc pass $i ($set) rc=$?"
done
python tools/pmc_mfma_summary.py $OUT/s19b_mfma $OUT/s19b_mfma.json > /dev/null
find $OUT/s19b_mfma -name "*.csv" -size +5M -delete
python - <<'PY'
import json
d = json.load(open('gpurun_out/s19b_mfma.json'))
rows = d if isinstance(d, list) else d.get('kernels', d)
for r in rows:
    if 'k_history_fused_x3' in r.get('kernel', '') or 'k_history_warp_vm' in r.get('kernel', ''):
        print(json.dumps(r))
PY
