#!/bin/bash
# The last GPU action of a round: the full `-m gpu` suite, exactly as the driver runs it, with the tree's identity
# recorded next to the summary.  Run ON the GPU box:   gpurun -- 'scripts/final_gate.sh r05 <git sha>'
# (the box has no .git: pass `git rev-parse HEAD` from the container; the source hash below identifies the tree either way).
# Output: gpurun_out/<round>_gpu_suite.txt -> copy to profiles/<round>_gpu_suite.txt and commit; no kernel commit after it.
ROUND=${1:-rXX}; SHA=${2:-unknown}
mkdir -p gpurun_out
OUT=gpurun_out/${ROUND}_gpu_suite.txt
SRC_HASH=$(cat $(ls heal_amd/csrc/*.hip heal_amd/csrc/*.h include/*.h heal_amd/*.py | sort) | sha256sum | cut -c1-16)
{
  echo "round: $ROUND   git HEAD (as passed in): $SHA   source hash (csrc + include + heal_amd/*.py): $SRC_HASH"
  echo "library: $(sha256sum heal_amd/lib/libheal_amd.so | cut -c1-16)   date: $(date -u +%FT%TZ)"
  echo "command: python -m pytest tests/ -x -q -m gpu"
} > $OUT
python -m pytest tests/ -x -q -m gpu --durations=15 > gpurun_out/${ROUND}_gpu_suite.log 2>&1
RC=$?
tail -25 gpurun_out/${ROUND}_gpu_suite.log >> $OUT
echo "exit code: $RC" >> $OUT
cat $OUT
exit $RC
