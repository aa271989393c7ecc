#!/bin/bash
# Round-3 session 2: DA pipelined sampler A/B (unit kernel / pipe 2 waves / pipe 3 waves), parity tests, whole GPU suite.
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
echo "== $(date)" > $OUT/box.txt
for v in "0 3" "1 2" "1 3"; do set -- $v
  FBBEV_DA_PIPE=$1 FBBEV_DA_PIPE_WPS=$2 timeout 300 python tools/time_fb.py BL2 4 30 4 > $OUT/fb_pipe$1_wps$2.json 2>$OUT/fb_pipe$1_wps$2.err; echo "time_fb pipe=$1 wps=$2 rc=$?" | tee -a $OUT/box.txt; cut -c1-400 $OUT/fb_pipe$1_wps$2.json
done
FBBEV_DA_PIPE=1 FBBEV_DA_PIPE_WPS=3 timeout 300 python tools/time_fb.py REF 4 30 1 > $OUT/fb_REF_pipe.json 2>&1; cut -c1-300 $OUT/fb_REF_pipe.json
FBBEV_DA_PIPE=0 timeout 300 python tools/time_fb.py REF 4 30 1 > $OUT/fb_REF_unit.json 2>&1; cut -c1-300 $OUT/fb_REF_unit.json
cd /tmp
for v in "0 3" "1 2" "1 3"; do set -- $v
  FBBEV_DA_PIPE=$1 FBBEV_DA_PIPE_WPS=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fb_pipe$1_wps$2 -- python $REPO/tools/time_fb.py BL2 4 20 4 > $OUT/prof_fb_pipe$1_wps$2.log 2>&1; echo "rocprof pipe=$1 wps=$2 rc=$?" | tee -a $OUT/box.txt
done
cd $REPO
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/box.txt; tail -15 $OUT/pytest_gpu.log | cut -c1-300
find $OUT -name "*kernel_trace.csv" -size +3M -delete
echo "== done $(date)" >> $OUT/box.txt
