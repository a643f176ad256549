#!/bin/bash
# Round profile: bench lines of every workload, rocprofv3 kernel stats of the default command, PMC passes.  GPU box:
#   gpurun -- 'bash scripts/profile_round.sh r02'
# Outputs land in gpurun_out/prof_<tag>/ ; copy the summaries to profiles/ afterwards.
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python bench.py > $OUT/bench_n1_scene5.json 2> $OUT/bench.err
tail -c 300 $OUT/bench_n1_scene5.json; echo
for w in single pair scene5_lidar; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline > $OUT/bench_n1_$w.json 2> $OUT/bench_$w.err
done
timeout 300 python bench.py --workload scene8_second_v2xvit --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_n1_scene8_second_v2xvit.json 2> $OUT/bench_scene8.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- \
    python bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats8 -- \
    python bench.py --workload scene8_second_v2xvit --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2> $OUT/rocprof8.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- \
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --eager --frames 1 > /dev/null 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- \
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --eager --frames 1 > /dev/null 2> $OUT/pmc_write.err
python scripts/pmc_summary.py $OUT/pmc_heal_kernels.txt $OUT/pmc_fetch $OUT/pmc_write --json $OUT/pmc_k2_traffic.json --agents 3
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_scene5.csv \;
find $OUT/stats8 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_scene8_second_v2xvit.csv \;
rm -rf $OUT/stats $OUT/stats8 $OUT/pmc_fetch $OUT/pmc_write   # keep the merge-back small
ls -la $OUT
