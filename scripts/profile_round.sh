#!/bin/bash
# Round profile: default bench line, rocprofv3 kernel stats of the same command, PMC passes. Run on the GPU box:
#   gpurun -- 'bash scripts/profile_round.sh r01'
# Outputs land in gpurun_out/prof_<tag>/ ; copy the summaries to profiles/ afterwards (scripts/collect_profiles.sh).
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python bench.py > $OUT/bench_n1_scene5.json 2> $OUT/bench.err
tail -c 600 $OUT/bench_n1_scene5.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- \
    python bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- \
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --eager > /dev/null 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- \
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --eager > /dev/null 2> $OUT/pmc_write.err
python scripts/pmc_summary.py $OUT/pmc_heal_kernels.txt $OUT/pmc_fetch $OUT/pmc_write --json $OUT/pmc_k2_traffic.json --agents 3
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_scene5.csv \;
rm -rf $OUT/stats/*/*kernel_trace.csv $OUT/pmc_fetch $OUT/pmc_write   # keep the merge-back small
ls -la $OUT
