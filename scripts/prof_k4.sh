#!/bin/bash
# rocprofv3 kernel stats of scripts/k4_bench.py (both K4 paths), K4 kernels only.  GPU box only.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_k4
rm -rf $OUT; mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python scripts/k4_bench.py > $OUT/log.txt 2>&1
python - <<'PY'
import csv, glob, re
pat = re.compile(r"k_lss|k_canvas|k_conv1x1|Radix|radix|k_sort|k_scan|fill|Memset|k_hist|k_scatter")
for f in glob.glob("gpurun_out/prof_k4/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat.search(r["Name"]):
            print(f"{r['Name'][:90]:<92}{r['Calls']:>6}{float(r['AverageNs'])/1e3:>9.1f}{float(r['MinNs'])/1e3:>9.1f}{float(r['MaxNs'])/1e3:>9.1f}")
PY
rm -f $OUT/*/*kernel_trace.csv
