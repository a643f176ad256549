# round 6, call F: restored stem v2 + new bench plumbing + 2-rank gloo self-proof on one GPU
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "pillar or pfn" 2>&1 | tail -3
python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "round6" 2>&1 | tail -3
python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r06/bench_f.json 2> gpurun_out/r06/bench_f.err; tail -3 gpurun_out/r06/bench_f.err
python -c "
import json; d=json.load(open('gpurun_out/r06/bench_f.json')); print(d['value'], d['ms_per_step'], d['serial']['ms_per_step']); r=d['roofline']; print(r['kernel'][:40], r['frac'], r.get('split'), r['traffic_source']); print([ (o['kernel'][:28], o['frac']) for o in d['roofline_other']])"
for wl in scene5 scene8_second_v2xvit; do
HEAL_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 --workload $wl > gpurun_out/r06/bench_gloo2_$wl.json 2> gpurun_out/r06/bench_gloo2_$wl.err; echo rc=$?; grep "\[bench\]" gpurun_out/r06/bench_gloo2_$wl.err | head -5
python -c "
import json,sys; d=json.loads(open('gpurun_out/r06/bench_gloo2_$wl.json').read().strip().splitlines()[-1]); print('$wl', d['value'], d['ms_per_step'], d['n_gpus'], d['config']['sharded_equals_single'], d['config']['job'].get('sharded_check'), d['config']['collective'])"
done
