#!/bin/bash
# Round profile: PMC traffic passes first (bench.py reads them for `roofline.traffic`), then the bench lines of every workload and
# the rocprofv3 kernel stats of the two BASELINE configurations.  GPU box:
#   gpurun -- 'bash scripts/profile_round.sh r04'
# Outputs land in gpurun_out/prof_<tag>/ ; copy the summaries to profiles/ afterwards (the PMC json is also written straight
# into profiles/ of the box's copy so that the bench lines of this same run carry it).
set -u
TAG=${1:-r04}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT profiles
export TMPDIR=/tmp
for W in scene5 scene8_second_v2xvit; do
  EXTRA=""; [ $W = scene8_second_v2xvit ] && EXTRA="--workload scene8_second_v2xvit"
  timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$W -- \
      python bench.py $EXTRA --steps 3 --warmup 1 --no-cpu-baseline --eager --frames 1 > /dev/null 2> $OUT/pmc_fetch_$W.err
  timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$W -- \
      python bench.py $EXTRA --steps 3 --warmup 1 --no-cpu-baseline --eager --frames 1 > /dev/null 2> $OUT/pmc_write_$W.err
  python scripts/pmc_summary.py $OUT/pmc_heal_kernels_$W.txt $OUT/pmc_fetch_$W $OUT/pmc_write_$W \
      --json $OUT/pmc_k2_traffic_$W.json --agents 3 --json-all $OUT/${TAG}_pmc_traffic_$W.json
  cp $OUT/${TAG}_pmc_traffic_$W.json profiles/${TAG}_pmc_traffic_$W.json
  rm -rf $OUT/pmc_fetch_$W $OUT/pmc_write_$W
done
cp $OUT/pmc_k2_traffic_scene5.json profiles/pmc_k2_traffic.json
timeout 600 python bench.py > $OUT/bench_n1_scene5.json 2> $OUT/bench.err
tail -c 300 $OUT/bench_n1_scene5.json; echo
timeout 600 python bench.py --workload single > $OUT/bench_n1_single.json 2> $OUT/bench_single.err    # BASELINE configs 1 / 2, with cpu_baseline
for w in pair scene5_lidar single_native pair_native; do     # (the *_native workloads: round 5, the YAMLs' 480 x 240 range)
  timeout 300 python bench.py --workload $w --no-cpu-baseline > $OUT/bench_n1_$w.json 2> $OUT/bench_$w.err
done
timeout 420 python bench.py --workload scene8_second_v2xvit --steps 10 --warmup 3 > $OUT/bench_n1_scene8_second_v2xvit.json 2> $OUT/bench_scene8.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- \
    python bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats8 -- \
    python bench.py --workload scene8_second_v2xvit --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2> $OUT/rocprof8.err
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_scene5.csv \;
find $OUT/stats8 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_scene8_second_v2xvit.csv \;
# sidecars: the library build these in-graph durations belong to (bench.py's `rocprof_in_graph_mean_us` checks them)
cp heal_amd/lib/libheal_amd.stamp $OUT/kernel_stats_scene5.stamp; cp heal_amd/lib/libheal_amd.stamp $OUT/kernel_stats_scene8_second_v2xvit.stamp
rm -rf $OUT/stats $OUT/stats8
[ "${FAST:-0}" = "1" ] && { ls -la $OUT; exit 0; }      # FAST=1: the round's lines and stats only (the micro-benchmarks below are unchanged kernels)
# per-kernel micro-benchmarks of the round's kernels
timeout 200 python scripts/k4_bench.py > $OUT/${TAG}_k4_bench.json 2> /dev/null
timeout 200 python scripts/k5_bench.py --json $OUT/${TAG}_k5_bench.json > /dev/null 2>&1
timeout 200 python scripts/c1t_bench.py --json $OUT/${TAG}_c1t_bench.json > /dev/null 2>&1
timeout 200 python scripts/conv_gemm_bench.py > $OUT/${TAG}_conv_gemm_bench.txt 2> /dev/null
timeout 100 python scripts/stem_bench.py > $OUT/${TAG}_stem_bench.txt 2> /dev/null
timeout 100 python scripts/k8_bench.py > $OUT/${TAG}_k8_bench.json 2> /dev/null
bash scripts/k8_prof.sh 2> /dev/null | grep "heal::" > $OUT/${TAG}_k8_kernels.txt
bash scripts/k3_rulebook_prof.sh > $OUT/${TAG}_k3_rulebook_kernels.txt 2> /dev/null
timeout 100 python scripts/linear_bench.py > $OUT/${TAG}_linear_bench.txt 2> /dev/null
bash scripts/k4_dbg.sh 0 2 16 64 > $OUT/${TAG}_k4_anatomy.txt 2>&1
timeout 100 python scripts/k4_stamps.py > $OUT/${TAG}_k4_stamps.txt 2> /dev/null
ls -la $OUT
