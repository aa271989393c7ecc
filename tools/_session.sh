OUT=gpurun_out/s16; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_backward_projection.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
timeout 200 python tools/time_train.py REF 4 2>&1 | tail -1 | tee $OUT/train_REF_B4.json
R=$(pwd); cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_def -- python $R/tools/time_train.py REF 4 > $R/$OUT/train_def.json 2> $R/$OUT/err.txt
f=$(ls $R/$OUT/prof_def/*/*kernel_stats.csv | head -1); grep "k_da_cross_attn_bwd\|k_da_bwd" $f | awk -F'",' '{print substr($1,1,50), $2,$3,$4}'
cd $R
find $OUT -name "*kernel_trace.csv" -delete
