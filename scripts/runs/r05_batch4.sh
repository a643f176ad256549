mkdir -p gpurun_out/r05
for d in 1 2 3 4; do
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --frames-in-flight $d 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('depth $d', d['value'], d['ms_per_step'], d.get('serial',{}).get('value'))"
done
