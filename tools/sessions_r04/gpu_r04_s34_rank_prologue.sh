#!/bin/bash
# segmented scatter with the hoisted pair loads and 16-byte column sums: parity, build time, per-kernel averages
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rank or sort or seg or index or interval or cache" -p no:cacheprovider 2>&1 | tail -2
for cfg in "BL2 16" "REF 16" "BL2 4" "REF 4" "BL2 16"; do
  timeout 120 python tools/time_rank.py $cfg 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["config"], d["B"], d["rank_build_ms"], d["checksum"][:2])'
done
cd /tmp; rm -rf $OUT/r04_prof_rank; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_prof_rank -- python $REPO/tools/time_rank.py BL2 16 > $OUT/r04_prof_rank.log 2>&1; cd $REPO
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r04_prof_rank/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
cp $(ls $OUT/r04_prof_rank/*/*kernel_stats.csv | head -1) $OUT/r04_rank_stats_after_prologue.csv
