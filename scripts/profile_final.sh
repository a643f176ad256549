#!/bin/bash
# Short re-run of the two BASELINE bench lines + rocprofv3 kernel stats after a late kernel change (the PMC traffic passes and the
# per-kernel micro-benchmarks of scripts/profile_round.sh are not repeated).   gpurun -- 'bash scripts/profile_final.sh r04'
set -u
TAG=${1:-r04}
OUT=$PWD/gpurun_out/prof_${TAG}_final
mkdir -p $OUT
export TMPDIR=/tmp
timeout 100 python bench.py > $OUT/bench_n1_scene5.json 2> $OUT/bench.err
tail -c 200 $OUT/bench_n1_scene5.json; echo
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- \
    python bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_scene5.csv \;
rm -rf $OUT/stats
timeout 100 python bench.py --workload scene8_second_v2xvit --steps 10 --warmup 3 > $OUT/bench_n1_scene8_second_v2xvit.json 2> $OUT/bench_scene8.err
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats8 -- \
    python bench.py --workload scene8_second_v2xvit --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2> $OUT/rocprof8.err
find $OUT/stats8 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_scene8_second_v2xvit.csv \;
rm -rf $OUT/stats8
timeout 40 python scripts/wino_bench.py $OUT/${TAG}_wino_bench.json > /dev/null 2>&1
ls $OUT
