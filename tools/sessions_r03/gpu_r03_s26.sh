#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for l in graph eager graph eager; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt-storage --launch $l > $OUT/bench_$l.json 2> $OUT/bench_$l.err || tail -5 $OUT/bench_$l.err
  python - $l <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/bench_{sys.argv[1]}.json'))
print(sys.argv[1], 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'gpu p50', d['step_gpu_ms_p10_p50_p90'][1], 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'equal', d.get('graph_step_equals_eager_step'))
PY
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_graph -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-storage > $OUT/prof_graph.log 2>&1; echo "rocprof rc=$?"
cd $REPO
python - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/prof_graph/**/*kernel_stats.csv',recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:10]: print(r['Name'][:80].ljust(80), r['Calls'], r['AverageNs'])
PY
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -2
find $OUT -name "*kernel_trace.csv" -size +3M -delete
