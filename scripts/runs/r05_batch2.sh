mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_models.py -q -m gpu -k "config2_3" 2>&1 | tail -3
python -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "not 200" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_scene5.json 2> gpurun_out/r05/bench_scene5.err; tail -c 600 gpurun_out/r05/bench_scene5.err
python -c "
import json; d=json.load(open('gpurun_out/r05/bench_scene5.json')); print(d['value'], d['ms_per_step'], d.get('serial',{}).get('value'), d['roofline']['frac'])"
for wl in scene5 scene8_second_v2xvit; do
HEAL_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 --workload $wl > gpurun_out/r05/bench_gloo2_$wl.json 2> gpurun_out/r05/bench_gloo2_$wl.err; echo rc=$?; tail -c 300 gpurun_out/r05/bench_gloo2_$wl.err
python -c "
import json,sys; d=json.load(open('gpurun_out/r05/bench_gloo2_$wl.json')); print('$wl', d['value'], d['ms_per_step'], d['n_gpus'], d['config']['parallelism'], d['config']['frames_in_flight'])"
done
