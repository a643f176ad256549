#!/bin/bash
# usage: bash scripts/pmc_probe.sh <which> ; prints mean counters per dispatch of the heal:: kernels
export TMPDIR=/tmp
W=${1:-conv1x1}
rm -rf /tmp/pmcp
timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmcp -- python scripts/pmc_probe.py $W > /dev/null 2> /tmp/pmcp.err
timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d /tmp/pmcp2 -- python scripts/pmc_probe.py $W > /dev/null 2>> /tmp/pmcp.err
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("/tmp/pmcp", "/tmp/pmcp2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "heal::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:<32} {sum(v)/len(v):>16.0f}")
PY
tail -3 /tmp/pmcp.err | cut -c1-200
