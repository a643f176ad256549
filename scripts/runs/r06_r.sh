OUT=gpurun_out/prof_r06; mkdir -p $OUT
for wl in scene5 scene8_second_v2xvit; do
  for coll in gather p2p; do
    HEAL_COLLECTIVE=$coll HEAL_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
      --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 --workload $wl 2> $OUT/bench_n2_${coll}_$wl.err | tail -n1 > $OUT/bench_n2_gloo_one_gpu_${coll}_$wl.json
    python -c "
import json; d=json.loads(open('$OUT/bench_n2_gloo_one_gpu_${coll}_$wl.json').read().strip().splitlines()[-1]); print('$wl $coll', d['value'], d['ms_per_step'], d['config']['sharded_equals_single'], d['config']['job']['sharded_check'])" || grep "\[bench\]" $OUT/bench_n2_${coll}_$wl.err | tail -2
  done
done
timeout 400 python scripts/scaling_model.py --workload scene5 --json $OUT/r06_scaling_model_scene5.json > $OUT/scaling_scene5.log 2>&1; tail -6 $OUT/scaling_scene5.log
timeout 400 python scripts/scaling_model.py --workload scene8_second_v2xvit --json $OUT/r06_scaling_model_scene8_second_v2xvit.json > $OUT/scaling_scene8.log 2>&1; tail -4 $OUT/scaling_scene8.log
