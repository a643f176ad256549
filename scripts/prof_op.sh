#!/bin/bash
# rocprofv3 kernel stats of scripts/kernel_bench.py, filtered to kernels matching $1 (regex).  GPU box only.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_op
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python scripts/kernel_bench.py --reps 10 > $OUT/log.txt 2>&1
python - "$1" <<'PY'
import csv, glob, sys, re
pat = re.compile(sys.argv[1])
for f in glob.glob("gpurun_out/prof_op/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat.search(r["Name"]):
            print(f"{r['Name'][:80]:<82}{r['Calls']:>6}{float(r['AverageNs'])/1e3:>9.1f}{float(r['MinNs'])/1e3:>9.1f}{float(r['MaxNs'])/1e3:>9.1f}")
PY
