#!/bin/bash
# rocprofv3 kernel stats of the default bench command -> top kernels by total time (per step)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/stats_bench
rm -rf $OUT; mkdir -p $OUT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python bench.py --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/err.txt
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/stats_bench/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"total kernel time {tot/1e6:.1f} ms")
    for r in rows[:45]:
        print(f"{r['Name'][:88]:<90}{r['Calls']:>6}{float(r['TotalDurationNs'])/1e6:>9.2f}{float(r['AverageNs'])/1e3:>9.1f}{float(r['Percentage']):>7.2f}")
PY
rm -f $OUT/*/*kernel_trace.csv
