OUT=gpurun_out/s18; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_backward_projection.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
for v in f32 bf16 f16; do timeout 200 python tools/time_fb.py BL2 4 20 4 $v 2>&1 | tail -1 | tee -a $OUT/fb_BL3_B4.jsonl | cut -c1-330; done
timeout 200 python tools/time_fb.py REF 4 20 1 bf16 2>&1 | tail -1 | tee -a $OUT/fb_REF_B4.jsonl | cut -c1-330
R=$(pwd); cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_fb -- python $R/tools/time_fb.py BL2 4 10 4 bf16 > $R/$OUT/prof_fb.log 2>&1
cd $R
f=$(ls $OUT/prof_fb/*/*kernel_stats.csv | head -1); head -8 $f | awk -F'",' '{print substr($1,1,70), $2,$3,$4}'
find $OUT -name "*kernel_trace.csv" -delete
