export TMPDIR=/tmp
timeout 1500 python tools/scope_table.py gpurun_out/scope_table.json 2>&1 | tail -30 | cut -c1-220
