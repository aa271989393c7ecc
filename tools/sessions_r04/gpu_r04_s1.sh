#!/bin/bash
# round 4, session 1: the new GPU tests (default DA kernel at kernel level, tolerance-mode pooling, ADVICE fixes), the full-size
# BackwardProjection statistics, baseline timings of HEAD, and the MFMA-busy passes VERDICT r3 item 7 asks for.
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward_projection.py tests/test_gpu_parity.py -m gpu -q -s --timeout 600 -p no:cacheprovider \
  -k "pipelined or tolerance or full_size or trainable or autocast or module_vs_oracle" > $OUT/r04_s1_pytest.log 2>&1; echo "pytest rc=$?"
grep -E "max\|err\||BackwardProjection full size|passed|failed|Error" $OUT/r04_s1_pytest.log | cut -c1-400 | tail -30
rm -f $OUT/r04_time_fb_base.jsonl
timeout 300 python tools/time_fb.py BL2 4 50 4 2>/dev/null >> $OUT/r04_time_fb_base.jsonl
timeout 300 python tools/time_fb.py REF 4 50 1 2>/dev/null >> $OUT/r04_time_fb_base.jsonl
cut -c1-600 $OUT/r04_time_fb_base.jsonl
timeout 300 python tools/time_history.py 400 400 16 1 f16 noref vm > $OUT/r04_time_hist_default.json 2>/dev/null; cut -c1-500 $OUT/r04_time_hist_default.json
bash tools/pmc_mfma.sh r04_depthnet -- python tools/time_depthnet.py 4 bf16_channels_last
bash tools/pmc_mfma.sh r04_fb -- python tools/time_fb.py BL2 4 5 4
bash tools/pmc_mfma.sh r04_hist -- python tools/time_history.py 400 400 16 1 f16 noref cx3 vm
echo "== done $(date)"
