OUT=gpurun_out/s21; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_history.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
(timeout 200 python tools/time_history.py 100 100 8 1 f32; timeout 200 python tools/time_history.py 100 100 8 1 f16 noref; timeout 300 python tools/time_history.py 200 200 16 4 f32 noref; timeout 300 python tools/time_history.py 400 400 16 1 f16 noref; timeout 300 python tools/time_history.py 400 400 16 1 f32 noref) 2>&1 | grep "^{" | tee $OUT/time_history.jsonl | cut -c1-330
