#!/bin/bash
# One replayed step of a workload, kernel by kernel in program order (rocprofv3 --kernel-trace + scripts/step_kernels.py).
#   scripts/step_trace.sh <workload> [out.txt]
W=${1:-scene8_second_v2xvit}
OUT=${2:-gpurun_out/step_${W}.txt}
export TMPDIR=/tmp
D=/tmp/step_trace_$W
rm -rf $D
HEAL_PARALLEL_MODALITIES=0 rocprofv3 --kernel-trace --output-format csv -d $D -- \
    python bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --frames-in-flight 1 > /dev/null 2>&1
python scripts/step_kernels.py $D > $OUT
wc -l $OUT
