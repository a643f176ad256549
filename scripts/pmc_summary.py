"""Aggregate rocprofv3 --pmc counter_collection CSVs into a per-kernel table (mean per dispatch).

usage: python scripts/pmc_summary.py OUT.txt DIR_FETCH DIR_WRITE [--json K2.json]
Each DIR is the -d directory of one `rocprofv3 --pmc <COUNTER> --output-format csv` pass (counters are collected
in separate passes: FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950, MI355X_MICROARCH.md PMC table)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d):
    """-> (counter name, {kernel: (dispatches, mean value)}, {kernel: mean value over its LARGEST-grid dispatches})"""
    acc = defaultdict(lambda: [0, 0.0])
    big = defaultdict(lambda: [0, 0, 0.0])  # grid, n, sum
    name = None
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                name = row["Counter_Name"]
                v = float(row["Counter_Value"])
                acc[k][0] += 1
                acc[k][1] += v
                g = int(row.get("Grid_Size", 0) or 0)
                if g > big[k][0]:
                    big[k] = [g, 0, 0.0]
                if g == big[k][0]:
                    big[k][1] += 1
                    big[k][2] += v
    return name, {k: (n, s / n) for k, (n, s) in acc.items()}, {k: s / max(n, 1) for k, (g, n, s) in big.items()}


def main():
    out, d_fetch, d_write = sys.argv[1:4]
    _, fetch, fetch_big = load(d_fetch)
    _, write, write_big = load(d_write)
    keys = sorted(k for k in set(fetch) | set(write) if "heal::" in k)
    lines = ["# mean per dispatch, counter units are KB (rocprofv3 FETCH_SIZE / WRITE_SIZE), separate passes",
             "# gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read -> x2 for 16 B/lane streams",
             f"{'kernel':<78}{'dispatches':>11}{'FETCH_KB':>12}{'WRITE_KB':>12}"]
    for k in keys:
        n = fetch.get(k, write.get(k))[0]
        lines.append(f"{k[:76]:<78}{n:>11}{fetch.get(k, (0, 0.0))[1]:>12.1f}{write.get(k, (0, 0.0))[1]:>12.1f}")
    open(out, "w").write("\n".join(lines) + "\n")
    if "--json" in sys.argv:
        # K2 = map memset + k_pfn + k_canvas per (collated) launch; FETCH x2 (wide coalesced reads), WRITE as reported.
        # k_canvas is shared with K4 (camera canvases are smaller): only its largest-grid dispatches are K2's.
        tot = 0.0
        parts = {}
        for k in keys:
            if "k_pfn" in k or "k_canvas" in k:
                b = (2.0 * fetch_big.get(k, 0.0) + write_big.get(k, 0.0)) * 1024.0
                parts[k.split("(")[0]] = b
                tot += b
        agents = int(sys.argv[sys.argv.index("--agents") + 1]) if "--agents" in sys.argv else 1
        json.dump({"k2_traffic_bytes_per_launch": tot, "agents_per_launch": agents, "parts": parts,
                   "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; bytes = 2*FETCH_SIZE"
                             " + WRITE_SIZE (KB->B), k_pfn + k_canvas per launch (the hipMemsetAsync of the "
                             "cell->pillar map is not a kernel dispatch and is not counted)"},
                  open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


    if "--json-all" in sys.argv:
        # every heal:: kernel: mean HBM bytes per dispatch = 2 * FETCH_SIZE + WRITE_SIZE (KB -> B; the guide's gfx950 correction:
        # FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read).  bench.py reads this file for `roofline.traffic`.
        allk = {}
        for k in keys:
            name = k.split("(")[0].replace("void ", "").strip()
            n = fetch.get(k, write.get(k))[0]
            allk[name] = {"dispatches": n, "fetch_KB": round(fetch.get(k, (0, 0.0))[1], 1),
                          "write_KB": round(write.get(k, (0, 0.0))[1], 1),
                          "bytes_per_dispatch": (2.0 * fetch.get(k, (0, 0.0))[1] + write.get(k, (0, 0.0))[1]) * 1024.0}
        # the build stamp of the library whose kernels were counted: bench.py refuses the summary for any other build
        stamp_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "heal_amd", "lib", "libheal_amd.stamp")
        stamp = open(stamp_path).read().strip() if os.path.exists(stamp_path) else None
        json.dump({"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; bytes = (2 FETCH + WRITE) KB",
                   "lib_stamp": stamp, "kernels": allk}, open(sys.argv[sys.argv.index("--json-all") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
