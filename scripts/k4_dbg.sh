#!/bin/bash
# per-variant kernel times of K4 (rocprofv3 kernel stats), GPU box only
export TMPDIR=/tmp
for d in "$@"; do
  rm -rf /tmp/k4dbg; HEAL_K4_DBG=$d timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k4dbg -- python scripts/k4_dbg.py > /dev/null 2>&1
  python - "$d" <<'PY'
import csv, glob, sys
for f in glob.glob("/tmp/k4dbg/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_lss" in r["Name"] or "k_bev_stem" in r["Name"]:
            print(f"dbg={sys.argv[1]:>3} {r['Name'][6:22]:<18} calls {r['Calls']:>4} avg {float(r['AverageNs'])/1e3:7.1f}  min {float(r['MinNs'])/1e3:7.1f}  max {float(r['MaxNs'])/1e3:7.1f}")
PY
done
