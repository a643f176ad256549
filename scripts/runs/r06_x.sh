#!/bin/bash
# round 6: what the camera-only local stage (the 8-GPU period of scene5) consists of
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r06
cd /tmp
for ag in 3 4; do
  rm -rf /tmp/camp$ag
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/camp$ag -- python $R/scripts/cam_stage_probe.py --agent $ag > $R/gpurun_out/r06/cam_stage_$ag.log 2>&1
  f=$(find /tmp/camp$ag -name "*kernel_stats.csv" | head -1)
  cp "$f" $R/gpurun_out/r06/cam_stage_${ag}_kernel_stats.csv
  grep "local stage" $R/gpurun_out/r06/cam_stage_$ag.log
done
