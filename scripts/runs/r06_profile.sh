# round-6 evidence in one gpurun call: PMC traffic (recorded with the library's build stamp), bench lines of every workload, rocprofv3 kernel
# stats of the two BASELINE configurations (+ stamp sidecars), a 200-step repeat line, functional N = 2 lines on one GPU, K1 / stem benches
FAST=1 bash scripts/profile_round.sh r06 > gpurun_out/prof_r06.log 2>&1
OUT=gpurun_out/prof_r06
# the in-graph durations belong in profiles/ BEFORE the final scene5 line is taken, so that the line can quote them (rocprof_in_graph_mean_us)
cp $OUT/kernel_stats_scene5.csv profiles/r06_kernel_stats_scene5.csv; cp $OUT/kernel_stats_scene5.stamp profiles/r06_kernel_stats_scene5.stamp
cp $OUT/kernel_stats_scene8_second_v2xvit.csv profiles/r06_kernel_stats_scene8_second_v2xvit.csv; cp $OUT/kernel_stats_scene8_second_v2xvit.stamp profiles/r06_kernel_stats_scene8_second_v2xvit.stamp
timeout 600 python bench.py > $OUT/bench_n1_scene5.json 2> $OUT/bench.err
timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_n1_scene5_200steps.json 2> $OUT/bench200.err
for wl in scene5 scene8_second_v2xvit; do
  for coll in gather p2p; do
    HEAL_COLLECTIVE=$coll HEAL_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
      --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 --workload $wl 2> $OUT/bench_n2_${coll}_$wl.err | tail -n1 > $OUT/bench_n2_gloo_one_gpu_${coll}_$wl.json
  done
done
timeout 120 python scripts/k1_bench.py > $OUT/r06_k1_bench.txt 2>&1
timeout 200 python scripts/pillar_stem_bench.py > $OUT/r06_pillar_stem_bench.json 2> /dev/null
timeout 200 python scripts/k4_bench.py > $OUT/r06_k4_bench.json 2> /dev/null
for f in $OUT/bench_n1_*.json $OUT/bench_n2_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], (d.get('serial') or {}).get('value'), (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('traffic'))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
done
timeout 400 python scripts/scaling_model.py --workload scene5 --json $OUT/r06_scaling_model_scene5.json > $OUT/scaling_scene5.log 2>&1
timeout 400 python scripts/scaling_model.py --workload scene8_second_v2xvit --json $OUT/r06_scaling_model_scene8_second_v2xvit.json > $OUT/scaling_scene8.log 2>&1
tail -3 $OUT/scaling_scene5.log
