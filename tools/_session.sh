OUT=gpurun_out/s22; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_backward_projection.py tests/test_gpu_full_model.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
for v in f32 bf16; do timeout 200 python tools/time_fb.py BL2 4 20 4 $v 2>&1 | tail -1 | tee -a $OUT/fb_BL3_B4.jsonl | cut -c1-330; done
timeout 200 python tools/time_fb.py REF 4 20 1 2>&1 | tail -1 | tee -a $OUT/fb_REF_B4.jsonl | cut -c1-330
