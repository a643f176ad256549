# round 6, call A: today's baseline + the agent-chunked (depth-first) ResNeXt stage walk (HEAL_STAGE_CHUNK_MB)
mkdir -p gpurun_out/r06
run() {  # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r06/bench_$name.json 2> gpurun_out/r06/bench_$name.err
  python -c "
import json; d=json.load(open('gpurun_out/r06/bench_$name.json')); print('$name', d['value'], d['ms_per_step'], d.get('serial',{}).get('ms_per_step'), d['roofline']['frac'])" || tail -5 gpurun_out/r06/bench_$name.err
}
run base HEAL_STAGE_CHUNK_MB=0
run chunk110 HEAL_STAGE_CHUNK_MB=110
run chunk220 HEAL_STAGE_CHUNK_MB=220
run chunk60 HEAL_STAGE_CHUNK_MB=60
run base2 HEAL_STAGE_CHUNK_MB=0
python -m pytest tests/test_gpu_models.py -q -m gpu -k "config4 or collab_small" 2>&1 | tail -3
HEAL_STAGE_CHUNK_MB=110 python -m pytest tests/test_gpu_models.py -q -m gpu -k "config4 or collab_small or config2_3" 2>&1 | tail -3
