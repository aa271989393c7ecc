#!/bin/bash
# round 6, session 2: the encoder layer as one autograd node -- gradient tests + step time, A/B against FBBEV_TRAIN_FUSED=0
REPO=$(pwd); OUT=$REPO/gpurun_out/s02; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward_projection.py -q -x -s -p no:cacheprovider -k "training or trainable or autocast or host_sync or write_once or fp64 or owned" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log | cut -c1-400
python tools/train_path.py BL2 4 4 --steps 20 --sites --checksum > $OUT/train.json 2> $OUT/err.log; echo "rc=$?"; tail -5 $OUT/err.log; cut -c1-500 $OUT/train.json
FBBEV_TRAIN_FUSED=0 python tools/train_path.py BL2 4 4 --steps 20 --checksum > $OUT/train_off.json 2> $OUT/err_off.log; cut -c1-400 $OUT/train_off.json
