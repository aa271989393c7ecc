#!/bin/bash
# round 5, session 12: where the one-kernel DA sampler's time goes -- kernel-trace averages with the timing diagnostics
# (FBBEV_DA_FUSED_DIAG: 1 = no samples, 2 = every sample reads token 0) and the launch-shape knobs
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TAG=${TAG:-s12}
for knobs in ${KNOBS:-X=0 FBBEV_DA_FUSED_DIAG=1 FBBEV_DA_FUSED_DIAG=2 FBBEV_DA_FUSED_DIAG=5 FBBEV_DA_FUSED_DIAG=9 FBBEV_DA_FUSED_DIAG=13 FBBEV_DA_FUSED_DIAG=8 FBBEV_DA_FUSED_DIAG=4}; do
  rm -rf $OUT/${TAG}_prof
  cd /tmp && env $knobs timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -- python $REPO/tools/time_fb.py BL2 4 20 4 ${DT:-f32} > $OUT/${TAG}_prof.log 2>&1
  cd $REPO
  python - $TAG "$knobs" <<'PY'
import csv, glob, sys
for f in glob.glob('gpurun_out/%s_prof/**/*kernel_stats.csv' % sys.argv[1], recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_da_cross_attn_fused' in r['Name']:
            print(sys.argv[2], '|', r['Name'][:60], r['Calls'], round(float(r['AverageNs']) / 1e3, 1))
PY
done
