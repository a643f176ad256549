"""Ordered kernel list of one replayed step from a rocprofv3 --kernel-trace CSV (run the bench with HEAL_PARALLEL_MODALITIES=0
so that the order is the program order):  python scripts/step_kernels.py <trace dir> [--anchor k_voxb_insert] > list.txt"""
import csv, glob, os, re, sys
d = sys.argv[1]
anchor = sys.argv[sys.argv.index("--anchor") + 1] if "--anchor" in sys.argv else "k_voxb_insert"
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if anchor in r[2]]
# a step is one of the LONG anchor-to-anchor intervals (set-up code voxelises too); among those take the third-shortest (a replay)
most = max(b - a for a, b in zip(starts[:-1], starts[1:]))
iv = sorted((max(r[1] for r in rows[a:b]) - rows[a][0], a, b) for a, b in zip(starts[:-1], starts[1:]) if b - a >= most // 2)
_, a, b = iv[min(2, len(iv) - 1)]
# start the listing at the frame load that precedes the anchor: walk back over copies / fills
while a > 0 and ("copyBuffer" in rows[a - 1][2] or "fillBuffer" in rows[a - 1][2] or "FillFunctor" in rows[a - 1][2]):
    a -= 1
t0 = rows[a][0]
for s, e, n in rows[a:b]:
    short = re.sub(r"\(.*", "", n)
    short = re.sub(r"void |at::native::|\(anonymous namespace\)::", "", short)[:90]
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f}  {short}")
