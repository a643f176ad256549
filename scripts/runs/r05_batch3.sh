mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_stress.py -x -q -m gpu -k "guard or fill" 2>&1 | tail -5
python -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "p2p" 2>&1 | tail -8
for wl in scene5; do
HEAL_COLLECTIVE=p2p HEAL_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 --workload $wl > gpurun_out/r05/bench_gloo2_p2p_$wl.json 2> gpurun_out/r05/bench_gloo2_p2p_$wl.err; echo rc=$?; tail -c 400 gpurun_out/r05/bench_gloo2_p2p_$wl.err
tail -n1 gpurun_out/r05/bench_gloo2_p2p_$wl.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$wl', d['value'], d['ms_per_step'], d['n_gpus'], d['config']['parallelism'], d['config']['frames_in_flight'], d.get('serial',{}).get('value'))"
done
