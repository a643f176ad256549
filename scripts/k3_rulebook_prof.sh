#!/bin/bash
# per-kernel averages of the K3 rulebook kernels inside the config-5 bench (rocprofv3 kernel trace of a short run)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p3
rocprofv3 --kernel-trace --stats -d /tmp/p3 --output-format csv -- python /root/repo/bench.py --workload scene8_second_v2xvit --steps 6 --warmup 2 --no-cpu-baseline --frames-in-flight 1 > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/p3/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows:
    n=r["Name"]
    if any(k in n for k in ("k_sp_","k_sort","k_scan","k_mean_vfe","k_vox")) and "k_sp_conv2" not in n:
        print(f"{n[:70]:70s} {r['Calls']:>6} {float(r['AverageNs'])/1e3:8.1f}")
PY
