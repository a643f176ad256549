#!/bin/bash
# round 6: soak -- long timed regions of the headline workloads (hangs / drift / rare failures), each under its own timeout
mkdir -p gpurun_out/r06
for spec in "scene5 3000" "scene8_second_v2xvit 400" "pair 3000" "single_native 5000"; do
  set -- $spec
  timeout 300 python bench.py --workload $1 --steps $2 --warmup 20 --no-cpu-baseline > gpurun_out/r06/soak_$1.json 2> gpurun_out/r06/soak_$1.err
  echo "$1 rc=$? $(python -c "import json,sys; d=json.loads(open('gpurun_out/r06/soak_$1.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['steps'])" 2>&1 | tail -1)"
done
