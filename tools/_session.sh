export TMPDIR=/tmp; mkdir -p gpurun_out/s19
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/s19/bench.json 2> gpurun_out/s19/err.txt; python -c "
import json;d=json.load(open('gpurun_out/s19/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac']);print(d.get('bf16_storage'))"; tail -3 gpurun_out/s19/err.txt
