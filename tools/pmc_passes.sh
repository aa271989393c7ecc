#!/bin/bash
# Several rocprofv3 --pmc passes of one command (each counter set in its OWN run, kernel trace only -- the combination
# rules of the GPU pool), summarised per kernel into gpurun_out/<tag>_pmc.json by tools/pmc_summary.py.
#   bash tools/pmc_passes.sh TAG -- python tools/time_fb.py BL2 4 5 4
TAG=$1; shift; shift
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd $REPO && timeout -k 5 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/${TAG}_pmc/p$i -- "$@" > $OUT/${TAG}_pmc_p$i.log 2>&1 )
  echo "pmc pass $i ($set) rc=$?"
done
cd $REPO
python tools/pmc_summary.py $OUT/${TAG}_pmc $OUT/${TAG}_pmc.json
find $OUT/${TAG}_pmc -name "*.csv" -size +5M -delete
