OUT=gpurun_out/s20; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "index or rank or lift or cache or graph" 2>&1 | tail -2
R=$(pwd); cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-storage > $R/$OUT/bench.json 2> $R/$OUT/err.txt
cd $R
f=$(ls $OUT/prof/*/*kernel_stats.csv | head -1); head -12 $f | awk -F'",' '{print substr($1,1,60), $2,$3,$4}'
find $OUT -name "*kernel_trace.csv" -delete
cut -c1-200 $OUT/bench.json
timeout 100 python tools/time_rank.py REF 16 2>&1 | tail -2 | cut -c1-300
