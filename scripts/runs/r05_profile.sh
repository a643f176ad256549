# round-5 evidence in one gpurun call: PMC traffic, bench lines of every workload, rocprofv3 kernel stats, functional N = 2 lines
FAST=1 bash scripts/profile_round.sh r05 > gpurun_out/prof_r05.log 2>&1
OUT=gpurun_out/prof_r05
for wl in scene5 scene8_second_v2xvit; do
  for coll in gather p2p; do
    HEAL_COLLECTIVE=$coll HEAL_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
      --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 --workload $wl 2> $OUT/bench_n2_${coll}_$wl.err | tail -n1 > $OUT/bench_n2_gloo_one_gpu_${coll}_$wl.json
  done
done
timeout 120 python scripts/k1_bench.py > $OUT/r05_k1_bench.txt 2>&1
for f in $OUT/bench_n1_*.json $OUT/bench_n2_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], (d.get('serial') or {}).get('value'), (d.get('roofline') or {}).get('frac'))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
done
