#!/bin/bash
# K1 kernels inside the scene-5 / config-5 bench (rocprofv3 kernel trace of a short run, one frame at a time)
cd /tmp && export TMPDIR=/tmp
for w in scene5 scene8_second_v2xvit; do
rm -rf /tmp/p1
rocprofv3 --kernel-trace --stats -d /tmp/p1 --output-format csv -- python /root/repo/bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --frames-in-flight 1 > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/p1/**/*kernel_stats.csv",recursive=True)[0]
print("$w")
for r in csv.DictReader(open(f)):
    if "k_vox" in r["Name"]: print(f"  {r['Name'][:60]:60s} {r['Calls']:>6} {float(r['AverageNs'])/1e3:8.1f}")
PY
done
