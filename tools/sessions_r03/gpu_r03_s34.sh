#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_backward_projection.py tests/test_gpu_full_model.py -m gpu -q -x --timeout 600 -p no:cacheprovider > $OUT/pytest_bp.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_bp.log | cut -c1-200
timeout 200 python tools/time_train.py BL2 4 4 2>/dev/null | cut -c1-200
