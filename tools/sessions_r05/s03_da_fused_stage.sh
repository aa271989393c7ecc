#!/bin/bash
# round 5, session 3: the fused DA sampler with the two coarse levels of the pyramid staged in LDS (wave-private staging region =
# the transposition tile + 1.9 KB): parity tests, A/B against FBBEV_DA_FUSED_STAGE=0, kernel stats, TCP / LDS counters
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward_projection.py tests/test_gpu_block_kernels.py -m gpu -q -s -x --timeout 600 -p no:cacheprovider > $OUT/s03_pytest.log 2>&1; echo "pytest rc=$?"
grep -o "\[observed\].*\|one-kernel DA.*" $OUT/s03_pytest.log | grep -i "DA\|da_cross" | head -12
tail -3 $OUT/s03_pytest.log | cut -c1-300
rm -f $OUT/s03_time_fb.jsonl
for rep in 1 2; do
for knobs in "" "FBBEV_DA_FUSED_STAGE=0" "FBBEV_DA_FUSED_STAGE=440"; do
  for cfg in "BL2 4 40 4" "REF 1 40 1"; do
    env $knobs timeout 300 python tools/time_fb.py $cfg 2>/dev/null | sed "s/^{/{\"knobs\": \"$knobs\", /" >> $OUT/s03_time_fb.jsonl
  done
done
done
python - <<'PY'
import json
for l in open('gpurun_out/s03_time_fb.jsonl'):
    d = json.loads(l); print(d['knobs'] or 'default', d['config'], d['B'], d['levels'], 'eager', round(d['ms_fb'], 4), 'graph', round(d['ms_fb_graph'], 4))
PY
cd /tmp
for tag in stage nostage; do
  rm -rf $OUT/s03_prof_$tag
  if [ $tag = nostage ]; then export FBBEV_DA_FUSED_STAGE=0; else unset FBBEV_DA_FUSED_STAGE; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s03_prof_$tag -- python $REPO/tools/time_fb.py BL2 4 30 4 > $OUT/s03_prof_$tag.log 2>&1; echo "rocprof $tag rc=$?"
done
unset FBBEV_DA_FUSED_STAGE
cd $REPO
python - <<'PY'
import csv, glob
for tag in ('stage', 'nostage'):
    f = glob.glob(f'gpurun_out/s03_prof_{tag}/**/*kernel_stats.csv', recursive=True)[0]
    for r in list(csv.DictReader(open(f)))[:6]:
        print(tag, r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
bash tools/pmc_passes.sh s03_stage -- python tools/time_fb.py BL2 4 5 4 > $OUT/s03_pmc_stage.log 2>&1
FBBEV_DA_FUSED_STAGE=0 bash tools/pmc_passes.sh s03_nostage -- python tools/time_fb.py BL2 4 5 4 > $OUT/s03_pmc_nostage.log 2>&1
python - <<'PY'
import json
for tag in ('stage', 'nostage'):
    d = json.load(open(f'gpurun_out/s03_{tag}_pmc.json'))
    for k, v in d.items():
        if 'k_da_cross_attn_fused' in k:
            print(tag, {x: v.get(x) for x in ('TCP_TOTAL_CACHE_ACCESSES_sum', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_LDS', 'SQ_LDS_BANK_CONFLICT', 'SQ_WAVE_CYCLES', 'frac_parked_waitcnt_barrier', 'frac_issue_stall', 'frac_issuing', 'FETCH_SIZE', 'WRITE_SIZE', 'L2_hit_rate')})
PY
find $OUT -name "*kernel_trace.csv" -size +20M -delete
