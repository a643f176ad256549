"""Timeline of one replayed step from a rocprofv3 --kernel-trace CSV: wall time, device-busy time (union of kernel intervals),
idle gaps, concurrency, and the kernels ranked by the wall time they own exclusively.
    python scripts/timeline.py <dir with *_kernel_trace.csv> [--anchor k_voxb_insert] [--out file]"""
import csv, glob, os, re, sys


def main():
    d = sys.argv[1]
    anchor = sys.argv[sys.argv.index("--anchor") + 1] if "--anchor" in sys.argv else "k_voxb_insert"
    out = open(sys.argv[sys.argv.index("--out") + 1], "w") if "--out" in sys.argv else sys.stdout
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"])[:60],
                         r.get("Queue_Id", "?")))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if anchor in r[2]]
    if len(starts) < 4:
        print("anchor not found often enough", len(starts), file=out); return
    # anchor-to-anchor intervals; the graph replays are the shortest ones (eager / instrumented passes are longer): take the
    # median of the five shortest
    iv = []
    for a, b in zip(starts[:-1], starts[1:]):
        w = max(r[1] for r in rows[a:b]) - rows[a][0]
        iv.append((w, a, b))
    print("intervals (us, kernels):", [(round(w / 1e3), b - a) for w, a, b in iv], file=out)
    _, a, b = sorted(iv)[min(2, len(iv) - 1)]
    step = rows[a:b]
    t0 = step[0][0]
    wall = max(r[1] for r in step) - t0
    ssum = sum(r[1] - r[0] for r in step)
    # union + exclusive ownership
    ev = sorted([(r[0], 1, i) for i, r in enumerate(step)] + [(r[1], -1, i) for i, r in enumerate(step)])
    active = set(); last = t0; busy = 0; own = {}; conc = {}
    gaps = []
    prev_end_kernel = None
    for t, kind, i in ev:
        dt = t - last
        if dt > 0:
            n = len(active)
            conc[n] = conc.get(n, 0) + dt
            if n:
                busy += dt
                for j in active:
                    own[step[j][2]] = own.get(step[j][2], 0) + dt / n
            else:
                gaps.append((dt, prev_end_kernel, None, last - t0))
        last = t
        if kind == 1:
            if not active and gaps and gaps[-1][2] is None:
                g = gaps[-1]; gaps[-1] = (g[0], g[1], step[i][2], g[3])
            active.add(i)
        else:
            active.discard(i); prev_end_kernel = step[i][2]
    print(f"step: {len(step)} kernels, wall {wall / 1e3:.1f} us, sum of durations {ssum / 1e3:.1f} us, busy (union) {busy / 1e3:.1f} us, "
          f"idle {(wall - busy) / 1e3:.1f} us in {len(gaps)} gaps", file=out)
    print("concurrency histogram (us):", {k: round(v / 1e3, 1) for k, v in sorted(conc.items())}, file=out)
    queues = {}
    for r in step:
        queues[r[3]] = queues.get(r[3], 0) + (r[1] - r[0])
    print("per-queue kernel time (us):", {k: round(v / 1e3, 1) for k, v in queues.items()}, file=out)
    print("largest idle gaps (us, after -> before, at):", file=out)
    for g in sorted(gaps, reverse=True)[:15]:
        print(f"   {g[0] / 1e3:6.1f}  {g[1]} -> {g[2]}  @{g[3] / 1e3:.0f}", file=out)
    small = sum(1 for g in gaps if g[0] < 3000)
    print(f"gaps < 3 us: {small}, total {sum(g[0] for g in gaps if g[0] < 3000) / 1e3:.1f} us; mean gap {sum(g[0] for g in gaps) / max(1, len(gaps)) / 1e3:.2f} us", file=out)
    print("wall-time ownership (us):", file=out)
    for k, v in sorted(own.items(), key=lambda kv: -kv[1])[:40]:
        cnt = sum(1 for r in step if r[2] == k)
        print(f"   {v / 1e3:8.1f}  x{cnt:<3d} {k}", file=out)
    # coarse phases: when does the last side-queue kernel end (join), etc.
    main_q = max(queues, key=queues.get)
    side_end = max((r[1] for r in step if r[3] != main_q), default=t0)
    # the main queue in launch order, runs of the same kernel merged: where the serial part of the step goes
    print("main-queue sequence (us at start: kernel x count = total us):", file=out)
    seq = [r for r in step if r[3] == main_q]
    i = 0
    while i < len(seq):
        j = i; tot = 0
        while j < len(seq) and seq[j][2] == seq[i][2]:
            tot += seq[j][1] - seq[j][0]; j += 1
        print(f"   @{(seq[i][0] - t0) / 1e3:7.0f}  {seq[i][2][:48]:48s} x{j - i:<3d} {tot / 1e3:7.1f}", file=out)
        i = j
    print(f"main queue {main_q}; last side-queue kernel ends at {(side_end - t0) / 1e3:.1f} us of {wall / 1e3:.1f}", file=out)


if __name__ == "__main__":
    main()
