# round 6, call B: the LiDAR pillar stem (heal_pfn_pillars + heal_pillar_stem_block) -- parity, kernel bench, scene A/B
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "pillar or pfn" 2>&1 | tail -5
python scripts/pillar_stem_bench.py > gpurun_out/r06/pillar_stem_bench.json 2> gpurun_out/r06/pillar_stem_bench.err; cat gpurun_out/r06/pillar_stem_bench.json; tail -3 gpurun_out/r06/pillar_stem_bench.err
python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "golden or small or config2_3 or config4" 2>&1 | tail -5
run() {  # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r06/bench_$name.json 2> gpurun_out/r06/bench_$name.err
  python -c "
import json; d=json.load(open('gpurun_out/r06/bench_$name.json')); print('$name', d['value'], d['ms_per_step'], d.get('serial',{}).get('ms_per_step'), d['roofline']['frac'])" || tail -5 gpurun_out/r06/bench_$name.err
}
run dense HEAL_K2_POOLED=0
run pillar HEAL_K2_POOLED=1
run dense2 HEAL_K2_POOLED=0
run pillar2 HEAL_K2_POOLED=1
for wl in single pair scene5_lidar; do
  HEAL_K2_POOLED=0 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --workload $wl 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl dense', d['value'], d['ms_per_step'])"
  HEAL_K2_POOLED=1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --workload $wl 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl pillar', d['value'], d['ms_per_step'])"
done
