mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "shared_k4 or round6 or heterogeneous or concurrent_modality or frames_in_flight" 2>&1 | tail -4
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "bev_pool or bev_stem or lss" 2>&1 | tail -3
run() {  # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r06/bench_$name.json 2> gpurun_out/r06/bench_$name.err
  python -c "
import json; d=json.load(open('gpurun_out/r06/bench_$name.json')); k4=[o for o in d['roofline_other'] if o['kernel'].startswith('K4')][0]; print('$name', d['value'], d['ms_per_step'], d.get('serial',{}).get('ms_per_step'), 'K4', k4['launch_ms'], k4['frac'], k4['launches'])" || tail -5 gpurun_out/r06/bench_$name.err
}
run n_multi0 HEAL_K4_MULTI=0
run n_multi1 HEAL_K4_MULTI=1
run n_multi0b HEAL_K4_MULTI=0
run n_multi1b HEAL_K4_MULTI=1
