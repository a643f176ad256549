# round 6, call E: pillar stem v2b (grouped gathers) + camera-crop pyramid walk
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "pillar or pfn" 2>&1 | tail -3
for d in 0 8; do
HEAL_PS_DBG=$d python scripts/pillar_stem_bench.py 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dbg $d', r['kernel_own_us'].get('pillar_stem_block'), r['kernel_own_us'].get('pfn_pillars'), round(r['pillar_chain_us'],1))"
done
python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "round6 or golden or small or config4 or heterogeneous" 2>&1 | tail -5
run() {  # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r06/bench_$name.json 2> gpurun_out/r06/bench_$name.err
  python -c "
import json; d=json.load(open('gpurun_out/r06/bench_$name.json')); print('$name', d['value'], d['ms_per_step'], d.get('serial',{}).get('ms_per_step'), d['roofline']['frac'])" || tail -5 gpurun_out/r06/bench_$name.err
}
run e_nocrop HEAL_PYRAMID_CAMCROP=0
run e_crop HEAL_PYRAMID_CAMCROP=1
run e_nocrop2 HEAL_PYRAMID_CAMCROP=0
run e_crop2 HEAL_PYRAMID_CAMCROP=1
