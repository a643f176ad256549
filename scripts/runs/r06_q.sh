mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "voxel" 2>&1 | tail -4
python -m pytest tests/test_gpu_stress.py -x -q -m gpu 2>&1 | tail -3
python scripts/k1_bench.py 2>&1 | tail -4
python -c "
import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r06/bench_q.json 2> gpurun_out/r06/bench_q.err; tail -2 gpurun_out/r06/bench_q.err
python -c "
import json; d=json.load(open('gpurun_out/r06/bench_q.json')); print(d['value'], d['ms_per_step'], d['serial']['ms_per_step']); print([ (o['kernel'][:24], o['launch_ms'], o['frac']) for o in d['roofline_other'] if o['kernel'].startswith(('K1','K2'))])"
