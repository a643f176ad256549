mkdir -p gpurun_out/r06
for d in 1 2 3; do for par in 1 0; do
HEAL_PARALLEL_MODALITIES=$par python bench.py --steps 40 --warmup 10 --no-cpu-baseline --frames-in-flight $d 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('depth $d par $par', d['value'], d['ms_per_step'], d.get('serial',{}).get('ms_per_step'))"
done; done
for q in 2 8; do
GPU_MAX_HW_QUEUES=$q python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hwq $q', d['value'], d['ms_per_step'], d.get('serial',{}).get('ms_per_step'))"
done
