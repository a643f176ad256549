#!/bin/bash
# usage: bash scripts/k3_pmc.sh [dbg list] ; SQ / LDS / TCP counters of k_sp_conv2 on the real 64->64 layer (mean per dispatch)
export TMPDIR=/tmp
cd /tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
L=${1:-0}
rm -rf /tmp/k3p1 /tmp/k3p2 /tmp/k3p3
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d /tmp/k3p1 -- python $R/scripts/k3_dbg.py $L > /dev/null 2> /tmp/k3p.err
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d /tmp/k3p2 -- python $R/scripts/k3_dbg.py $L > /dev/null 2>> /tmp/k3p.err
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS_SMEM --output-format csv -d /tmp/k3p3 -- python $R/scripts/k3_dbg.py $L > /dev/null 2>> /tmp/k3p.err
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("/tmp/k3p1", "/tmp/k3p2", "/tmp/k3p3"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_sp_conv2" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:<32} {sum(v)/len(v):>16.0f}   n={len(v)}")
PY
tail -3 /tmp/k3p.err | cut -c1-200
