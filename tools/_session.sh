mkdir -p gpurun_out/s17
timeout 120 ./tools/micro/lds_atomic_bench | tee gpurun_out/s17/lds_atomic_bench.jsonl
