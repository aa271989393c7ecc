#!/bin/bash
# training path: Z-mean / re-add backward through fbbev_volume_zreduce -- parity, training step
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -k "zreduce or train or module or view_transform or fb_view" -p no:cacheprovider 2>&1 | tail -3
echo "BL2: $(python tools/time_train.py BL2 4 4 2>/dev/null | tail -1 | cut -c1-330)"
echo "REF: $(python tools/time_train.py REF 4 1 2>/dev/null | tail -1 | cut -c1-330)"
