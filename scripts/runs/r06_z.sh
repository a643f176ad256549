#!/bin/bash
# round 6: timeline of ONE frame at a time (serial latency): what runs alone on the chip
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r06
cd /tmp; rm -rf /tmp/ser
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ser -- python $R/bench.py --frames-in-flight 1 --steps 6 --warmup 2 --no-cpu-baseline > /tmp/ser.json 2> /tmp/ser.err
f=$(find /tmp/ser -name "*kernel_trace.csv" | head -1)
cp "$f" $R/gpurun_out/r06/serial_kernel_trace.csv
python $R/scripts/serial_timeline.py "$f" | tee $R/gpurun_out/r06/serial_timeline.txt
tail -c 400 /tmp/ser.json
