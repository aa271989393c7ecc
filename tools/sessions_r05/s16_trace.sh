#!/bin/bash
# round 5, session 16: kernel trace of S3 at configs[2] (and the shipped shape) with the current build + the block / projection GPU tests
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_block_kernels.py tests/test_gpu_backward_projection.py tests/test_gpu_parity.py -m gpu -q --timeout 900 -p no:cacheprovider -x > $OUT/s16_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/s16_pytest.log | cut -c1-200
for cfg in "BL2 4 20 4" "REF 1 20 1"; do
  rm -rf $OUT/s16_prof
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s16_prof -- python $REPO/tools/time_fb.py $cfg > $OUT/s16_prof.log 2>&1
  cd $REPO
  tail -1 $OUT/s16_prof.log | cut -c1-300
  python - "$cfg" <<'PY'
import csv, glob, sys
for f in glob.glob('gpurun_out/s16_prof/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:16]:
        print(sys.argv[1], '|', r['Name'][:70], r['Calls'], round(float(r['AverageNs']) / 1e3, 1))
PY
done
