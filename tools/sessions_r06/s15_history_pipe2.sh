#!/bin/bash
# round 6, session 15: pipelined history step with the convolutions in 256-thread workgroups (warp waves share the CU) -- sweep + overlap trace
REPO=$(pwd); OUT=$REPO/gpurun_out/s15; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_history.py -q -x -p no:cacheprovider 2>&1 | tail -2
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], "pipe", d["pipelined_step"], "chunks", d["chunks"], "fused_ms", d["fused_ms"], "warp_ms", d["warp_ms"])'
run() { timeout 600 python tools/time_history.py 400 400 16 1 f16 noref cx3 vm 2>>$OUT/err1.log | tee -a $OUT/hist.jsonl | python -c "$P" "$1"; }
HIST_PIPE=0 run "two-kernels"
HIST_PIPE=0 FBBEV_HX3_WAVES=4 run "two-kernels,conv-256-threads"
for nw in 4 8; do for tu in 4 2; do for ch in 1 4 10 25; do
  HIST_CHUNKS=$ch FBBEV_HISTORY_STEP_WAVES=$nw FBBEV_HISTORY_VM_TU=$tu run "step_waves=$nw,warp_TU=$tu"
done; done; done
cd /tmp && HIST_CHUNKS=10 timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -- python $REPO/tools/time_history.py 400 400 16 1 f16 noref cx3 vm > $OUT/prof.log 2>&1
cd $REPO
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/s15/prof/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_history_warp_vm' in r['Kernel_Name'] or 'k_history_conv_bf16x3' in r['Kernel_Name']]
rows = rows[-40:]
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    print(r['Kernel_Name'][5:30], 'queue', r.get('Queue_Id'), 'start_us', (int(r['Start_Timestamp']) - t0) / 1e3, 'end_us', (int(r['End_Timestamp']) - t0) / 1e3)
PY
