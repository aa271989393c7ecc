OUT=gpurun_out/s14; mkdir -p $OUT; export TMPDIR=/tmp
for a in "--pipeline graphs" "--pipeline graphs --streams 3" "--pipeline graphs --streams 1" "--pipeline alternate --steps 100"; do
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline $a > $OUT/b.json 2> $OUT/err.txt; python -c "
import json;d=json.load(open('$OUT/b.json'));print('$a',d['value'],d['ms_per_step'],d['roofline']['frac'],{k:v for k,v in (d.get('pipelined') or {}).items() if k!='what'})"
done
tail -5 $OUT/err.txt
