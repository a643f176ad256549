#!/bin/bash
# round 6: SQ / LDS / VMEM counters of k_sp_tiles on the three thin layers of config 5 (mean per dispatch); three --pmc passes
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
rm -rf /tmp/k3t1 /tmp/k3t2 /tmp/k3t3
CMD="python $R/scripts/k3_bench.py --layers 3 --iters 5 --modes tiles --brief"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d /tmp/k3t1 -- $CMD > /dev/null 2> /tmp/k3t.err
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d /tmp/k3t2 -- $CMD > /dev/null 2>> /tmp/k3t.err
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC --output-format csv -d /tmp/k3t3 -- $CMD > /dev/null 2>> /tmp/k3t.err
python - <<'PY' | tee $R/gpurun_out/r06/k3_tiles_pmc.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("/tmp/k3t1", "/tmp/k3t2", "/tmp/k3t3"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_sp_tiles<" in r["Kernel_Name"] or "k_sp_nbr_tiles" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:<32} {sum(v)/len(v):>16.0f}   n={len(v)}")
PY
tail -3 /tmp/k3t.err | cut -c1-200
