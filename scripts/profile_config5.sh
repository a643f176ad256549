#!/bin/bash
# Config-5 part of the round profile (after a change that only touches the V2X-ViT path): bench line with CPU baseline, rocprofv3
# kernel stats, the linear / split-attention micro-benchmarks and the 1 -> 8 GPU model.   gpurun -- 'bash scripts/profile_config5.sh r04'
set -u
TAG=${1:-r04}
OUT=$PWD/gpurun_out/prof_${TAG}_c5
mkdir -p $OUT
export TMPDIR=/tmp
timeout 420 python bench.py --workload scene8_second_v2xvit --steps 10 --warmup 3 > $OUT/bench_n1_scene8_second_v2xvit.json 2> $OUT/bench_scene8.err
tail -c 400 $OUT/bench_n1_scene8_second_v2xvit.json; echo
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats8 -- \
    python bench.py --workload scene8_second_v2xvit --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2> $OUT/rocprof8.err
find $OUT/stats8 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_scene8_second_v2xvit.csv \;
rm -rf $OUT/stats8
timeout 100 python scripts/linear_bench.py > $OUT/${TAG}_linear_bench.txt 2> /dev/null
timeout 100 python scripts/split_bench.py > $OUT/${TAG}_split_bench.txt 2> /dev/null
timeout 400 python scripts/scaling_model.py --workload scene8_second_v2xvit --json $OUT/${TAG}_scaling_model_scene8_second_v2xvit.json > $OUT/scaling.log 2>&1
tail -6 $OUT/scaling.log
ls -la $OUT
