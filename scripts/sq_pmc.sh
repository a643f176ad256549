#!/bin/bash
# SQ / GRBM counters per kernel (mean per dispatch, grouped by kernel name and grid size) for any command:
#   bash scripts/sq_pmc.sh <out.txt> <command ...>
# One pass: GRBM_GUI_ACTIVE (kernel duration in shader cycles: effective clock = this / wall time), wave cycles split into
# parked (WAIT_ANY) / issue-stalled (WAIT_INST_ANY) / issuing (ACTIVE_INST_ANY), MFMA pipe busy cycles, LDS issue stalls.
export TMPDIR=/tmp
OUT=$1; shift
D=/tmp/sqpmc_$$
rm -rf $D
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \
    SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $D -- "$@" > /dev/null 2> $D.err
python - $D $OUT <<'PY'
import csv, glob, collections, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "heal::" not in n:
            continue
        n = re.sub(r"\(.*", "", n).replace("void ", "")
        acc[(n[:48], int(r.get("Grid_Size", 0) or 0))][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES",
        "SQ_BUSY_CYCLES", "SQ_WAIT_INST_LDS"]
with open(sys.argv[2], "w") as o:
    o.write("# mean per dispatch; SQ wave counters are quad-cycles summed over waves; parked/stall/issue = share of SQ_WAVE_CYCLES\n")
    o.write(f"{'kernel':<50}{'grid':>10}{'n':>5}{'gui_cycles':>12}{'parked':>8}{'stall':>7}{'issue':>7}{'lds_st':>7}{'mfma_busy/gui':>14}\n")
    for (k, g), cs in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
        m = {c: (sum(cs[c]) / len(cs[c]) if cs.get(c) else 0.0) for c in cols}
        wc = max(m["SQ_WAVE_CYCLES"], 1.0)
        o.write(f"{k:<50}{g:>10}{len(cs['GRBM_GUI_ACTIVE']):>5}{m['GRBM_GUI_ACTIVE']:>12.0f}{m['SQ_WAIT_ANY'] / wc:>8.2f}"
                f"{m['SQ_WAIT_INST_ANY'] / wc:>7.2f}{m['SQ_ACTIVE_INST_ANY'] / wc:>7.2f}{m['SQ_WAIT_INST_LDS'] / wc:>7.2f}"
                f"{m['SQ_VALU_MFMA_BUSY_CYCLES'] / max(m['GRBM_GUI_ACTIVE'], 1.0):>14.2f}\n")
PY
tail -2 $D.err | cut -c1-200
wc -l $OUT
