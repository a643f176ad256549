# round 6, call C: pillar stem v2 (pixel-compacted) -- parity, kernel bench, scene A/B
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "pillar or pfn" 2>&1 | tail -5
python scripts/pillar_stem_bench.py > gpurun_out/r06/pillar_stem_bench2.json 2> gpurun_out/r06/pillar_stem_bench2.err; cat gpurun_out/r06/pillar_stem_bench2.json; tail -3 gpurun_out/r06/pillar_stem_bench2.err
python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "golden or small or config2_3 or config4" 2>&1 | tail -5
run() {  # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r06/bench_$name.json 2> gpurun_out/r06/bench_$name.err
  python -c "
import json; d=json.load(open('gpurun_out/r06/bench_$name.json')); print('$name', d['value'], d['ms_per_step'], d.get('serial',{}).get('ms_per_step'), d['roofline']['frac'])" || tail -5 gpurun_out/r06/bench_$name.err
}
run c_dense HEAL_K2_POOLED=0
run c_v1 HEAL_PILLAR_STEM=1
run c_v2 HEAL_PILLAR_STEM=2
run c_dense2 HEAL_K2_POOLED=0
run c_v2b HEAL_PILLAR_STEM=2
