#!/bin/bash
# MFMA-busy evidence for one command: three rocprofv3 --pmc passes, each counter set in its OWN run with the
# kernel trace only, summarised per kernel by tools/pmc_mfma_summary.py into gpurun_out/<tag>_mfma.json.
#   bash tools/pmc_mfma.sh TAG -- python tools/time_depthnet.py 4 bf16only
TAG=$1; shift; shift
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" \
           "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  ( cd $REPO && timeout -k 5 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/${TAG}_mfma/p$i -- "$@" > $OUT/${TAG}_mfma_p$i.log 2>&1 )
  echo "mfma pmc pass $i ($set) rc=$?"
done
cd $REPO
python tools/pmc_mfma_summary.py $OUT/${TAG}_mfma $OUT/${TAG}_mfma.json
find $OUT/${TAG}_mfma -name "*.csv" -size +5M -delete
