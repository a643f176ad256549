mkdir -p gpurun_out/r06
for wl in scene5 scene8_second_v2xvit; do
HEAL_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 --workload $wl > gpurun_out/r06/bench_gloo2_$wl.json 2> gpurun_out/r06/bench_gloo2_$wl.err; echo rc=$?; grep "\[bench\]" gpurun_out/r06/bench_gloo2_$wl.err | head -5
python -c "
import json,sys; d=json.loads(open('gpurun_out/r06/bench_gloo2_$wl.json').read().strip().splitlines()[-1]); print('$wl', d['value'], d['ms_per_step'], d['n_gpus'], d['config']['sharded_equals_single'], d['config']['job'].get('sharded_check'), d['config']['collective'])"
done
