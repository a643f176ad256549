#!/bin/bash
# Effective shader clock and MFMA-pipe utilisation per kernel dispatch: GRBM_GUI_ACTIVE (cycles, summed over the 8 XCDs) and
# SQ_VALU_MFMA_BUSY_CYCLES together with the dispatch durations of the kernel trace of the SAME run.
#   bash scripts/clock_pmc.sh <out.txt> <command ...>
export TMPDIR=/tmp
OUT=$1; shift
D=/tmp/clkpmc_$$
rm -rf $D
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $D -- "$@" > /dev/null 2> $D.err
python - $D $OUT <<'PY'
import csv, glob, collections, re, sys
dur = {}
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "heal::" not in n or r["Dispatch_Id"] not in dur:
            continue
        n = re.sub(r"\(.*", "", n).replace("void ", "")
        k = (n[:44], int(r.get("Grid_Size", 0) or 0))
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            acc[k]["ns"].append(dur[r["Dispatch_Id"]][0])
with open(sys.argv[2], "w") as o:
    o.write("# mean per dispatch (under the counter pass: durations are longer than un-profiled ones); clock = GRBM_GUI_ACTIVE / 8 XCDs / duration;\n"
            "# mfma = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs ... reported as busy / gui / 128)\n")
    o.write(f"{'kernel':<46}{'grid':>10}{'n':>5}{'us':>10}{'GHz':>8}{'mfma_util':>11}\n")
    for (k, g), cs in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("ns", [0]))):
        if not cs.get("ns"):
            continue
        gui = sum(cs["GRBM_GUI_ACTIVE"]) / len(cs["GRBM_GUI_ACTIVE"])
        ns = sum(cs["ns"]) / len(cs["ns"])
        mf = sum(cs.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])) / max(len(cs.get("SQ_VALU_MFMA_BUSY_CYCLES", [])), 1)
        o.write(f"{k:<46}{g:>10}{len(cs['ns']):>5}{ns / 1e3:>10.1f}{gui / 8 / ns:>8.2f}{mf / max(gui, 1) / 128:>11.2f}\n")
PY
tail -2 $D.err | cut -c1-160
cat $OUT | head -30
