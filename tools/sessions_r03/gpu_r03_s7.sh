#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
FBBEV_TRAIN_PROFILE_COPIES=1 FBBEV_TRAIN_PROFILE=$OUT/r03_train_step_kernels_bf16_clbn.json timeout 900 python bench.py --mode train --steps 2 --warmup 2 > $OUT/train_clbn.json 2> $OUT/train_clbn.err; echo "train rc=$?"; cut -c1-200 $OUT/train_clbn.json; tail -3 $OUT/train_clbn.err
for v in "0 0" "1 0" "1 1"; do set -- $v
  FBBEV_DA_PIPE=$1 FBBEV_DA_PATCH=$2 timeout 300 python tools/time_fb.py BL2 4 30 4 > $OUT/fb_pipe$1_patch$2.json 2>/dev/null; echo "pipe=$1 patch=$2: $(cut -c1-260 $OUT/fb_pipe$1_patch$2.json)"
done
FBBEV_DA_PATCH=1 timeout 300 python tools/time_fb.py REF 4 30 1 > $OUT/fb_REF_patch1.json 2>/dev/null; cut -c1-260 $OUT/fb_REF_patch1.json
FBBEV_DA_PATCH=0 timeout 300 python tools/time_fb.py REF 4 30 1 > $OUT/fb_REF_patch0.json 2>/dev/null; cut -c1-260 $OUT/fb_REF_patch0.json
cd /tmp; export TMPDIR=/tmp
FBBEV_DA_PATCH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fb_patch1 -- python $REPO/tools/time_fb.py BL2 4 20 4 > $OUT/prof_fb_patch1.log 2>&1
cd $REPO
timeout 900 python -m pytest tests/test_gpu_backward_projection.py -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_bp.log 2>&1; echo "pytest bp rc=$?"; tail -3 $OUT/pytest_bp.log | cut -c1-300
find $OUT -name "*kernel_trace.csv" -size +3M -delete
