#!/bin/bash
# rocprofv3 kernel stats of the default bench command -> gpurun_out/<tag>_kernel_stats.csv (+ the bench line under rocprof)
# usage: bash scripts/stats_bench.sh <tag> [bench args...]
export TMPDIR=/tmp
TAG=${1:-stats}; shift
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python bench.py --no-cpu-baseline "$@" > $OUT/bench_under_rocprof.json 2> $OUT/err.txt
find $OUT -name "*kernel_stats.csv" -exec cp {} $PWD/gpurun_out/${TAG}_kernel_stats.csv \;
cp $OUT/bench_under_rocprof.json $PWD/gpurun_out/${TAG}_bench_under_rocprof.json
rm -rf $OUT
python - "$PWD/gpurun_out/${TAG}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over the run")
for r in rows[:28]:
    print(f"{r['Name'][:70]:<72}{r['Calls']:>6}{float(r['TotalDurationNs'])/1e6:>9.2f} ms{float(r['Percentage']):>7.2f}%{float(r['AverageNs'])/1e3:>9.1f} us")
PY
