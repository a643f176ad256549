python -m pytest tests/test_gpu_kernels.py -x -q -k "nms or quad_iou or decode" 2>&1 | tail -3; python scripts/k8_bench.py 2>&1 | tail -1; cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/p8 --output-format csv -- python /root/repo/scripts/k8_bench.py > /dev/null 2>&1; python - <<PY
import csv,glob
f=glob.glob("/tmp/p8/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("nms","decode","topk","rank")): print(r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3)
PY
