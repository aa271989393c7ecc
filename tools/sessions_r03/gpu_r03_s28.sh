#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_backward_projection.py -m gpu -q --timeout 600 -p no:cacheprovider -k "rows_linear" 2>&1 | tail -4
