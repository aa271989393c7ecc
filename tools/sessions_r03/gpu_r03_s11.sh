#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
rm -f $OUT/r03_exp_history_occupancy.jsonl
for v in "0 0" "80 0" "160 0" "0 90" "80 90"; do set -- $v
  FBBEV_HISTORY_WARP_LDS_PAD_KB=$1 FBBEV_HISTORY_CONV_LDS_PAD_KB=$2 timeout 300 python tools/time_history.py 400 400 16 1 f16 noref cbf16 vm 2>/dev/null | sed "s/^{/{\"warp_lds_pad_kb\": $1, \"conv_lds_pad_kb\": $2, /" >> $OUT/r03_exp_history_occupancy.jsonl
done
cut -c1-330 $OUT/r03_exp_history_occupancy.jsonl
