export TMPDIR=/tmp; mkdir -p gpurun_out/s19
timeout 400 python tools/sweep_pool.py BL2 16 bf16 2>&1 > gpurun_out/s19/sweep_bf16_b.jsonl
timeout 400 python tools/sweep_pool.py BL2 16 f32 2>&1 > gpurun_out/s19/sweep_f32_b.jsonl
for f in gpurun_out/s19/sweep_bf16_b.jsonl gpurun_out/s19/sweep_f32_b.jsonl; do echo $f; grep variant $f | awk -F'"ms": ' '{print $2, $1}' | sort -n | head -8; done
